"""GPU: BASELINE config 4 - the debias edit on SDXL (uce_sd_debias.py:39-46,240-242): the SDXL-base architecture
(2.57 G-parameter U-Net, 140 attn2 to_k/to_v projections with 2048-d context from two text encoders) built by this
runtime, the FULL 166 400 x 2048 weight slab edited through the C ABI and compared with an fp64 evaluation of the
closed form on the same slab, and the SDXL sampling path (Euler, micro-conditioning) the debias loop calls."""
import numpy as np
import pytest
import torch

from oracle import uce_oracle as O

pytestmark = pytest.mark.gpu

PROFESSIONS = ["doctor", "nurse", "teacher", "engineer", "lawyer", "chef", "pilot", "farmer", "artist", "scientist",
               "firefighter", "police officer", "carpenter", "plumber", "electrician", "dentist", "pharmacist", "architect",
               "accountant", "librarian", "journalist", "photographer", "musician", "actor", "athlete", "banker", "cashier",
               "cleaner", "driver", "mechanic", "receptionist", "secretary", "soldier", "surgeon", "tailor", "waiter"]


@pytest.fixture(scope="module")
def sdxl():
    from uce_amd.sd import pipeline as sdp
    pipe = sdp.load_pipeline("stabilityai/stable-diffusion-xl-base-1.0", torch.float32, "cuda:0", synthetic=True, vae=True)
    yield pipe
    del pipe
    torch.cuda.empty_cache()


def test_sdxl_architecture_has_the_reference_edit_surface(sdxl):
    from uce_amd import edit as E, synth
    mods = E.collect_uce_modules(sdxl.unet)
    table = synth.sdxl_module_table()
    assert len(mods) == 140 and [n for n, _ in mods] == [n for n, _ in table]
    assert [tuple(m.weight.shape) for _, m in mods] == [(o, 2048) for _, o in table]
    assert sum(p.numel() for p in sdxl.unet.parameters()) == 2_567_463_684          # SDXL-base U-Net
    pe = sdxl.encode_prompt(prompt="a doctor", device="cuda:0", num_images_per_prompt=1, do_classifier_free_guidance=False)
    assert pe[0].shape == (1, 77, 2048) and pe[2].shape == (1, 1280)


def test_sdxl_debias_full_slab_matches_fp64(sdxl, tmp_path):
    """debias.UCE on the real-size SDXL surface: 36 professions x (male, female), two scripted drift iterations
    (the reference's sampling is unseeded), every one of the 140 edited projections vs the fp64 closed form."""
    from safetensors.torch import load_file
    from tests import fakepipe
    from uce_amd import debias, edit as E
    # the U-Net (the 140 real-size modules) with CLIP-like concept embeddings: the randomly initialised text encoders of
    # the synthetic pipeline map every prompt to nearly the same vector (a singular system, not a test of the solver)
    pipe = fakepipe.FakePipe(sdxl.unet, 2048)
    rng = np.random.Generator(np.random.PCG64(4))
    scripted = [rng.uniform(-0.3, 0.3, size=(36, 2)) for _ in range(2)] + [np.zeros((36, 2))]
    it = iter(scripted)
    mods = E.collect_uce_modules(sdxl.unet)
    W_old = E.WeightSlab.from_modules(mods, "cuda:0")
    slab, path = debias.UCE(pipe, None, PROFESSIONS, ["male", "female"], [], 1.0, 1.0, 0.5, str(tmp_path), "sdxl_deb",
                            0.05, 0.1, 10, 20, 7.5, desired_ratios=[0.5, 0.5], max_iterations=5, device="cuda:0",
                            ratios_fn=lambda **kw: next(it))
    assert slab.data.shape == (166400, 2048)
    # fp64 closed form with the cumulative drift (uce_sd_debias.py:122-140): G = C_e + (sum_t D_t) C_debias
    emb = E.last_token_embeddings(pipe, PROFESSIONS + ["male", "female"], "cuda:0")
    C = torch.stack([emb[p] for p in PROFESSIONS]).double()
    Cd = torch.stack([emb["male"], emb["female"]]).double()
    Dsum = torch.from_numpy(scripted[0] + scripted[1]).cuda()
    G = C + Dsum @ Cd
    A = 0.5 * torch.eye(2048, dtype=torch.float64, device="cuda") + C.T @ C
    Delta = (G - C).T @ C @ torch.linalg.inv(A)                       # [d, d]
    worst = 0.0
    for lo in range(0, 166400, 16640):                                 # fp64 in row blocks (2.7 GB otherwise)
        w = W_old.data[lo:lo + 16640].double()
        want = w + w @ Delta
        worst = max(worst, O.rel_fro(slab.data[lo:lo + 16640], want))
    assert worst < 1e-5, worst
    state = load_file(path)
    assert len(state) == 140 and all(v.dtype == torch.float32 for v in state.values())
    name, m = mods[77]
    assert torch.equal(state[name + ".weight"], slab.views()[77].cpu())


def test_sdxl_sampling_path_runs_at_full_size(sdxl):
    """`pipe(concept, num_images_per_prompt=n, ...)` of the debias loop (uce_sd_debias.py:22-26) at SDXL's native 1024 x 1024:
    Euler steps, pooled-text + size/crop conditioning, bf16, cross-attention through uce_xattn_fwd (dh = 64)."""
    sdxl.to("cuda:0", torch.bfloat16)
    try:
        out = sdxl("a doctor", num_inference_steps=2, num_images_per_prompt=2, guidance_scale=7.5,
                   generator=torch.Generator().manual_seed(3))
        assert out.latents.shape == (2, 4, 128, 128) and bool(torch.isfinite(out.latents.float()).all())
        assert len(out.images) == 2 and out.images[0].size == (1024, 1024)
    finally:
        sdxl.to("cuda:0", torch.float32)
