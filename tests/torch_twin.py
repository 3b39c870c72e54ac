"""The all-torch twin of a 16-bit GPU model, for comparison runs: the product modules (uce_amd.sd.unet / pipeline) send 16-bit
GPU tensors to the hand-written kernels unconditionally - there is no switch in the product.  A test that wants the SAME weights
through torch's own ops patches the dispatch predicates here, for the duration of a `with` block."""
import contextlib


@contextlib.contextmanager
def torch_ops(*predicates: str):
    """Force the named predicates of uce_amd.sd.unet to False (default: `hip16`, the root one - every kernel off)."""
    from uce_amd.sd import unet as U
    names = predicates or ("hip16",)
    saved = {n: getattr(U, n) for n in names}
    try:
        for n in names:
            setattr(U, n, lambda *a, **k: False)
        yield
    finally:
        for n, f in saved.items():
            setattr(U, n, f)
