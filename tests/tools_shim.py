"""The three-slab module table the golden fixtures were generated on (tools/make_golden.py)."""
FIXTURE_TABLE = [
    ("down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k", 32),
    ("up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_v", 32),
    ("mid_block.attentions.0.transformer_blocks.0.attn2.to_k", 32),
]
