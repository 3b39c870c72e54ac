"""Where the embedding stage of a 1 500-concept UCE() goes (fp32 pipeline as the CLI loads it): tokenizer, text-encoder forward on the
prefix positions, gather - and the encoder's kernels by family (torch profiler)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E
from uce_amd.sd import pipeline as sdp

dev = torch.device("cuda:0")
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.float32, dev, synthetic=True, vae=False)
prompts = [f"artist number {i}" for i in range(1000)] + [f"kept artist {i}" for i in range(500)] + ["art"]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    emb = E.last_token_embeddings(pipe, prompts, dev, batch_size=None)
    torch.cuda.synchronize(); print("last_token_embeddings %.1f ms" % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter()
tok = pipe.tokenizer(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt")
print("tokenizer %.1f ms" % (1e3 * (time.perf_counter() - t0)))
idx = tok["attention_mask"].sum(1) - 2
n_pos = int(idx.max()) + 1
print("n_pos", n_pos)
ids = tok["input_ids"][:, :n_pos].to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        hs = pipe.text_encoder(input_ids=ids[:1024])[0]
        hs2 = pipe.text_encoder(input_ids=ids[1024:])[0]
    torch.cuda.synchronize(); print("encoder forward (1024 + %d strings) %.1f ms" % (len(prompts) - 1024, 1e3 * (time.perf_counter() - t0)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    with torch.no_grad():
        hs = pipe.text_encoder(input_ids=ids[:1024])[0]
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:14]
tot = sum(e.device_time_total for e in prof.key_averages())
print("device time total %.1f ms" % (tot / 1e3))
for e in rows:
    print("%8.2f ms x%4d  %s" % (e.device_time_total / 1e3, e.count, e.key[:100]))
