"""CPU: the evidence chain of the counter figures - tools/pmc_fold.py's attribution of rocprofv3 dispatches to the shapes of a
`bench.py --only xattn|sattn` pass, and a sanity gate over the committed profiles/traffic.json (a folded entry whose HBM bytes are
far BELOW what the launch has to move is a mis-attribution, not a fast kernel: round 5 shipped 4.3 MB for a 684 MB launch)."""
import csv
import importlib.util
import json
import os

import pytest

from uce_amd import REPO_ROOT


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def fold():
    return _load("pmc_fold", "tools/pmc_fold.py")


@pytest.fixture(scope="module")
def bench():
    return _load("bench_mod", "bench.py")


def _write_pass(path, rows, counter):
    """rows: [(kernel name, grid, value)] in dispatch order -> a rocprofv3 counter_collection CSV (two instances per dispatch)."""
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id",
                    "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                    "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for i, (name, grid, val) in enumerate(rows, 1):
            for half in (0.25, 0.75):
                w.writerow([i, i, 0, 0, 1, 1, grid, 7, name, 256, 0, 0, 64, 0, 32, counter, val * half, 0, 1])


def _line(shapes):
    return json.dumps({"kernel": "k", "shapes": shapes})


def test_fold_attributes_by_the_launch_counts_of_the_bench_line(fold, tmp_path):
    # two shapes, 8 calls each (4 untimed + 4 timed - what bench.time_kernel issues at iters = 4), the second shape on another
    # template form; torch's own kernels in between are ignored
    shapes = [{"B": 128, "Lq": 4096, "dh": 40, "launches": 8}, {"B": 128, "Lq": 64, "dh": 160, "launches": 8}]
    rows = [("void at::native::vectorized_elementwise_kernel<4>(int)", 1024, 5.0)]
    rows += [("void (anonymous namespace)::k_xattn_g<40, 16>(unsigned short const*, int)", 65536, 1000.0)] * 8
    rows += [("void at::native::vectorized_elementwise_kernel<4>(int)", 1024, 5.0)]
    rows += [("void (anonymous namespace)::k_xattn<160, 4>(unsigned short const*, int)", 2048, 10.0)] * 8
    p = tmp_path / "xattn_pmc_fetch_counter_collection.csv"
    _write_pass(p, rows, "FETCH_SIZE")
    g = fold.by_manifest(str(p), "xattn", shapes)
    assert sorted(g) == [("B128_Lq4096_dh40", "k_xattn"), ("B128_Lq64_dh160", "k_xattn")]
    assert [c["FETCH_SIZE"] for c in g[("B128_Lq4096_dh40", "k_xattn")]] == [1000.0] * 8      # the two instances summed
    assert [c["FETCH_SIZE"] for c in g[("B128_Lq64_dh160", "k_xattn")]] == [10.0] * 8


def test_fold_refuses_a_pass_that_does_not_hold_the_announced_launches(fold, tmp_path):
    # the round-5 defect: the line says 5 per shape, the pass holds 8
    shapes = [{"B": 128, "Lq": 4096, "dh": 40, "launches": 5}, {"B": 128, "Lq": 64, "dh": 160, "launches": 5}]
    rows = [("k_xattn_g<40, 16>(int)", 65536, 1000.0)] * 8 + [("k_xattn<160, 4>(int)", 2048, 10.0)] * 8
    p = tmp_path / "xattn_pmc_fetch_counter_collection.csv"
    _write_pass(p, rows, "FETCH_SIZE")
    with pytest.raises(fold.FoldError, match="16 launches .* announces 10"):
        fold.by_manifest(str(p), "xattn", shapes)


def test_fold_refuses_mixed_kernels_inside_one_shape(fold, tmp_path):
    # right total, wrong split: 6 + 10 launches attributed 8 + 8 puts two kernels / grids under the first shape
    shapes = [{"B": 128, "Lq": 4096, "dh": 40, "launches": 8}, {"B": 128, "Lq": 64, "dh": 160, "launches": 8}]
    rows = [("k_xattn_g<40, 16>(int)", 65536, 1000.0)] * 6 + [("k_xattn<160, 4>(int)", 2048, 10.0)] * 10
    p = tmp_path / "xattn_pmc_fetch_counter_collection.csv"
    _write_pass(p, rows, "FETCH_SIZE")
    with pytest.raises(fold.FoldError, match="not one kernel at one grid"):
        fold.by_manifest(str(p), "xattn", shapes)


def test_fold_separates_shapes_that_share_one_kernel_name_and_keeps_helpers_with_their_call(fold, tmp_path):
    # sattn: L = 256 and L = 64 run the same instantiation at different grids; only the first shape launches the k_vt helper
    shapes = [{"B": 128, "L": 1024, "dh": 80, "launches": 2}, {"B": 128, "L": 256, "dh": 160, "launches": 2},
              {"B": 128, "L": 64, "dh": 160, "launches": 2}]
    rows = []
    for _ in range(2):
        rows += [("k_vt(unsigned short const*)", 512, 3.0), ("k_sattn<80, true, 2, false>(int)", 4096, 100.0)]
    rows += [("k_sattn_p<160, true, true>(int)", 1024, 20.0)] * 2 + [("k_sattn_p<160, true, true>(int)", 256, 2.0)] * 2
    p = tmp_path / "sattn_pmc_write_counter_collection.csv"
    _write_pass(p, rows, "WRITE_SIZE")
    g = fold.by_manifest(str(p), "sattn", shapes)
    assert [c["WRITE_SIZE"] for c in g[("B128_L1024_dh80", "k_vt")]] == [3.0, 3.0]
    assert [c["WRITE_SIZE"] for c in g[("B128_L1024_dh80", "k_sattn")]] == [100.0, 100.0]
    assert [c["WRITE_SIZE"] for c in g[("B128_L256_dh160", "k_sattn")]] == [20.0, 20.0]
    assert [c["WRITE_SIZE"] for c in g[("B128_L64_dh160", "k_sattn")]] == [2.0, 2.0]
    assert ("B128_L256_dh160", "k_vt") not in g


def test_bench_line_is_read_from_the_pass_log_and_needs_launch_counts(fold, tmp_path):
    log = tmp_path / "xattn_pmc_fetch.log"
    log.write_text("[bench] noise\n" + _line([{"B": 2, "Lq": 64, "dh": 160, "launches": 8}]) + "\n")
    assert fold.bench_line(str(log))[0]["launches"] == 8
    log.write_text(_line([{"B": 2, "Lq": 64, "dh": 160}]) + "\n")
    with pytest.raises(fold.FoldError, match="no `launches`"):
        fold.bench_line(str(log))


def test_time_kernel_launch_count_has_one_source(bench):
    # the `launches` the --only lines announce is what time_kernel really issues
    calls = []
    import torch
    real = (torch.cuda.synchronize, torch.cuda.Event)

    class Ev:
        def __init__(self, enable_timing=False):
            pass

        def record(self):
            pass

        def elapsed_time(self, other):
            return 1.0

    torch.cuda.synchronize, torch.cuda.Event = (lambda *a, **k: None), Ev
    try:
        for iters, burst in ((4, None), (100, None), (300, None), (4, 0), (10, 3)):
            calls.clear()
            bench.time_kernel(lambda: calls.append(1), iters, burst)
            assert len(calls) == bench.kernel_launches(iters, burst)
    finally:
        torch.cuda.synchronize, torch.cuda.Event = real
    assert bench.kernel_launches(4) == 8 and bench.kernel_launches(300) == 400


# ---------------------------------------------------------------------------------------------------------------------------------
# profiles/traffic.json against the algorithmic bytes bench.py prices the same launches with

L2_RESIDENT_BYTES = 48e6      # a working set below this can live in the 8 x 4 MB L2s between back-to-back launches: no lower bound


def _traffic():
    return json.load(open(os.path.join(REPO_ROOT, "profiles", "traffic.json")))


def test_traffic_attention_entries_move_at_least_their_algorithmic_bytes(bench):
    t = _traffic()
    checked = 0
    for key, ent in t.get("xattn", {}).items():
        B, Lq, dh = (int(x.lstrip("BLqdh")) for x in key.split("_"))
        alg = bench.xattn_algorithmic_bytes(B, Lq, 8 * dh)
        assert ent["total_bytes"] <= 4.0 * alg + 2e6, (key, ent["total_bytes"], alg)
        if alg >= L2_RESIDENT_BYTES:
            assert ent["total_bytes"] >= 0.9 * alg, (key, ent["total_bytes"], alg)
            checked += 1
    for key, ent in t.get("sattn", {}).items():
        B, L, dh = (int(x.lstrip("BLdh")) for x in key.split("_"))
        alg = bench.sattn_algorithmic_bytes(B, L, 8 * dh)
        assert ent["total_bytes"] <= 4.0 * alg + 2e6, (key, ent["total_bytes"], alg)
        if alg >= L2_RESIDENT_BYTES:
            assert ent["total_bytes"] >= 0.9 * alg, (key, ent["total_bytes"], alg)
            checked += 1
    assert checked >= 6          # the generation-batch shapes of both kernels


def test_traffic_edit_entries_move_at_least_their_algorithmic_bytes(bench):
    t = _traffic()
    for name, (n_e, n_p, d, table, _) in bench.WORKLOADS.items():
        ent = t.get(name)
        if not ent:
            continue
        from uce_amd import synth
        rows = sum(o for _, o in (synth.sd14_module_table() if table == "sd14" else synth.sdxl_module_table()))
        N = n_e + n_p
        path = bench.edit_path(N, n_e, d, rows, 0)
        for kname, e in ent.items():
            if "total_bytes" not in e or "members" in e:
                continue
            bound, work, _peak, _unit, _note = bench.kernel_model(kname, path, N, n_e, d, rows)
            if bound == "hbm" and work >= L2_RESIDENT_BYTES:
                assert e["total_bytes"] >= 0.9 * work, (name, kname, e["total_bytes"], work)
        # the step: whatever the kernels, W comes in and goes out once
        step = [e for k, e in ent.items() if "total_bytes" in e and "members" not in e]
        q = bench.step_floor(path, N, n_e, d, rows)[0]
        assert sum(e.get("chain_bytes", e["total_bytes"]) for e in step) >= 0.9 * q, (name, q)
