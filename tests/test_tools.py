"""CPU: the profile-analysis tooling (scripts/overlap_report.py) on a hand-made kernel trace with known overlaps."""
import importlib.util
import io
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("overlap_report", os.path.join(ROOT, "scripts", "overlap_report.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_overlap_report_interval_arithmetic(tmp_path):
    t = _tool()
    ms = 1_000_000
    # queue 1 / stream 1: two GEMMs [0,4) [5,9); queue 2 / stream 2: a conv [2,7); queue 2 / stream 3: AdamW [8,10); idle nowhere but [4,5) on q1
    rows = [("g256::kernel<a>", 1, 1, 0, 4), ("g256::kernel<b>", 1, 1, 5, 9), ("cslab::conv_slab_kernel(x)", 2, 2, 2, 7),
            ("adamw_kernel(y)", 2, 3, 8, 10)]
    p = tmp_path / "t_kernel_trace.csv"
    with open(p, "w") as f:
        f.write('"Kind","Agent_Id","Queue_Id","Stream_Id","Kernel_Name","Start_Timestamp","End_Timestamp"\n')
        for name, q, st, s, e in rows:
            f.write(f'"KERNEL_DISPATCH","Agent 2",{q},{st},"{name}",{s * ms},{e * ms}\n')
    out = io.StringIO()
    r = t.report(t.load(str(p)), out=out)
    assert abs(r["span_ms"] - 10.0) < 1e-9 and r["idle_ms"] == 0.0
    fam = r["families"]
    assert abs(fam["gemm"]["kernel_ms"] - 8.0) < 1e-9 and abs(fam["gemm"]["wall_ms"] - 8.0) < 1e-9
    assert abs(fam["gemm"]["overlapped_ms"] - 5.0) < 1e-9      # conv over [2,4) and [5,7), AdamW over [8,9)
    assert abs(fam["conv"]["overlapped_ms"] - 4.0) < 1e-9 and abs(fam["adamw"]["overlapped_ms"] - 1.0) < 1e-9
    assert r["shared_queues"] == {"2": ["2", "3"]}             # two streams multiplexed onto hardware queue 2
    hist = t.depth_histogram(t.load(str(p)))
    assert hist[1] == 5 * ms and hist[2] == 5 * ms and hist[0] == 0
    assert t.family("void ffn_mid_bwd_kernel<unsigned short, 3, true>") == "rows" and t.family("ncclDevKernel_AllReduce") == "collective"
    # --last-ms keeps only the tail of the trace
    assert len(t.load(str(p), last_ms=2.5)) == 1


def test_attention2_lane_arithmetic_on_cpu():
    """scripts/exp/attn2_emulate.py restates the address formulas of csrc/attention2.hip lane by lane (the row permutation pi, both LDS
    read patterns, the C/D register <-> row map of the 32 x 32 x 16 MFMA, the v_permlane32_swap store packing, the lse / dsum hand-over
    order) on the documented instruction semantics and checks forward and fused backward of one head against float64 attention - and
    that both read patterns are bank-conflict free under the guide's LDS model.  S = 257 (one real row in the 9th block) and 240 (keys
    masked inside the 8th block); the GPU tests (test_one_tile_attention_blocks32) are the parity tests proper."""
    import subprocess
    import sys
    for S in ("257", "240"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp", "attn2_emulate.py"), S], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1500:]
        assert "OK" in out.stdout and "worst multiplicity of a 16-byte slot inside a service group = 1" in out.stdout
        assert "worst multiplicity of a bank inside a 32-lane group = 1" in out.stdout


def test_bf16x3_operand_image_cache_host_logic(monkeypatch):
    """CPU: ops.X3Images (one operand image per tensor and training step) and ops.Planes (a planes-only operand) without a GPU - the split
    is replaced by a counting stand-in.  A tensor is split once however many products read it; a rewritten tensor (version bump) is
    split again; producer-written planes are found without a split; backward-made images live in a short LRU; clear() forgets all;
    outside a step nothing is cached and planes-only results are refused."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
    from muse import ops
    calls = []

    def fake_split(t):
        calls.append(t)
        return torch.zeros((2,) + tuple(t.shape), dtype=torch.bfloat16)
    monkeypatch.setattr(ops, "_split_planes_now", fake_split)
    monkeypatch.setattr(ops, "require_gpu", lambda *a: None)
    a, b = torch.zeros(8, 16), torch.zeros(8, 16)
    ops.split_planes(a); ops.split_planes(a)
    assert len(calls) == 2                                        # no step running: every product splits its own operands
    im = ops.X3Images(recent=2)
    with ops.f32_gemms_as_bf16x3(True, im):
        p1 = ops.split_planes(a)
        assert ops.split_planes(a) is p1 and len(calls) == 3 and (im.hits, im.misses) == (1, 1)
        assert ops.split_planes(b) is not p1 and len(calls) == 4  # same shape, other storage: its own image
        ops._touched(a)                                           # a kernel rewrote a in place
        assert ops.split_planes(a) is not p1 and len(calls) == 5
        made = torch.ones((2, 8, 16), dtype=torch.bfloat16)
        c = torch.zeros(8, 16)
        im.put_planes(c, made)                                    # a producer kernel wrote c's planes itself
        assert ops.split_planes(c) is made and len(calls) == 5 and im.produced == 1
        with ops.f32_gemms_as_bf16x3(False):                      # (the products' own inner scope keeps the step's cache visible)
            assert ops.split_planes(c) is made
        im.backward = True
        g = [torch.zeros(8, 16) for _ in range(3)]
        for t in g:
            ops.split_planes(t)
        assert len(calls) == 8 and len(im.lru) == 2               # the oldest backward image was evicted ...
        ops.split_planes(g[0])
        assert len(calls) == 9                                    # ... and is split again when asked for
        assert ops.split_planes(c) is made                        # forward images stay until the step ends
        im.clear()
        assert not im.persist and not im.lru
    assert ops._X3_IMAGES[0] is None and not ops.planes_only_ok(512, 512)
    pl = ops.Planes(torch.zeros((2, 256, 128), dtype=torch.bfloat16))
    assert tuple(pl.shape) == (256, 128) and pl.dtype == torch.float32 and pl.stride() == (128, 1) and pl.stride(0) == 128
    assert pl.dim() == 2 and pl.numel() == 256 * 128 and pl.is_contiguous() and ops.split_planes(pl) is pl.planes and len(calls) == 9


def test_f16_mode_host_logic(monkeypatch):
    """CPU: the host side of the "f16" compute mode without a GPU - the cast is a counting stand-in.  One half image per tensor, scale and
    step; a gradient scale must be a power of two (it is undone exactly in alpha); backward images live in a short LRU, and so do a
    tape-less forward's (keep = False: no activation outlives its consumers); a producer's image is found without a cast and carries
    the pass's scale; the default gradient scale is 2^10 x the loss's token rows rounded up to a power of two; the dynamic policy halves
    on overflow (skip the step) and doubles after `growth_interval` good steps."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
    from muse import ops, tape_ops
    from muse._hip import MuseHipError
    casts = []

    def fake_cast(t, scale=1.0, stats=None):
        casts.append((t, scale))
        out = torch.zeros(t.shape, dtype=torch.float16)
        out._muse_scale = float(scale)
        return out
    monkeypatch.setattr(ops, "cast_to_f16", fake_cast)
    monkeypatch.setattr(ops.F16Images, "_ensure_stats", lambda self, device=None: None)
    im = ops.F16Images(recent=2)
    with pytest.raises(MuseHipError):
        im.set_grad_scale(3.0)
    with pytest.raises(MuseHipError):
        im.set_grad_scale(0.0)
    im.set_grad_scale(2.0 ** 19)
    a, b = torch.zeros(8, 16), torch.zeros(8, 16)
    i1 = im.image(a, 1.0)
    assert im.image(a, 1.0) is i1 and len(casts) == 1 and (im.hits, im.misses) == (1, 1)
    assert im.image(a, 2.0 ** 19) is not i1 and len(casts) == 2          # the same tensor as a gradient operand: another image
    assert im.image(b, 1.0) is not i1 and len(casts) == 3
    ops._touched(a)
    assert im.image(a, 1.0) is not i1 and len(casts) == 4                 # rewritten in place: converted again
    w16 = torch.zeros(8, 16, dtype=torch.float16)
    assert im.image(w16, 1.0) is w16 and len(casts) == 4                  # a weight's cached half copy is its own image
    im.backward = True
    made = torch.zeros((1, 8, 16), dtype=torch.float16)
    c = torch.zeros(8, 16)
    im.put_planes(c, made)                                                # a backward producer wrote c's image with the pass's scale
    got = im.image(c, im.grad_scale)
    assert got._muse_scale == 2.0 ** 19 and got.data_ptr() == made.data_ptr() and len(casts) == 4
    for t in [torch.zeros(8, 16) for _ in range(3)]:
        im.image(t, im.grad_scale)
    assert len(im.lru) == 2 and len(im.persist) == 4
    im.clear()
    im.backward, im.keep = False, False                                   # inference forward: nothing persists
    im.image(a, 1.0)
    assert not im.persist and len(im.lru) == 1
    # a planes-only half operand knows its scale
    ops._F16_IMAGES[0] = im
    try:
        im.backward = True
        pl = ops.Planes(torch.zeros((1, 256, 128), dtype=torch.float16))
        assert pl.half and pl.scale == 2.0 ** 19 and tuple(pl.shape) == (256, 128) and pl.half_image()._muse_scale == 2.0 ** 19
        assert im.image(pl, 1.0).data_ptr() == pl.planes.data_ptr()
        with pytest.raises(MuseHipError):
            im.image(ops.Planes(torch.zeros((2, 256, 128), dtype=torch.bfloat16)), 1.0)
    finally:
        ops._F16_IMAGES[0] = None

    class Host(tape_ops.TapeOps):
        pass
    h = Host()
    assert h.f16_grad_scale_for(16384) == 2.0 ** 24 and h.f16_grad_scale_for(16385) == 2.0 ** 25 and h.f16_grad_scale_for(1) == 1024.0
    h.__dict__["_loss_rows"] = 512
    stats = [(3, 0), (0, 7), (0, 0), (0, 0)]
    monkeypatch.setattr(Host, "f16_stats", lambda self, reset=True: stats.pop(0))
    assert h.f16_update_grad_scale(growth_interval=2) is False and h.f16_grad_scale == 2.0 ** 18      # overflow: halve, skip the step
    assert h.f16_update_grad_scale(growth_interval=2) is True and h.f16_grad_scale == 2.0 ** 18
    assert h.f16_update_grad_scale(growth_interval=2) is True and h.f16_grad_scale == 2.0 ** 19       # two good steps: double
    assert h.f16_update_grad_scale(growth_interval=2) is True and h.f16_grad_scale == 2.0 ** 19


def test_bf16x3_attention_shape_logic_and_step_timeline_on_cpu(tmp_path):
    """host logic added in round 6: which sequence shapes the bf16x3 attention takes in one tile / block by block / not at all
    (ops.attention_x3_supported / attention_x3_blocked), and scripts/step_timeline.py on a synthetic kernel trace (two steps of three
    streams: the per-bin busy shares add up)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
    from muse import ops
    for sq, skv, ok, blocked in [(256, 256, True, False), (256, 77, True, False), (256, 96, True, False), (256, 128, False, False),
                                 (1024, 1024, True, True), (1024, 77, True, True), (512, 512, True, True), (512, 300, False, True),
                                 (384, 384, False, True), (128, 128, False, False)]:
        assert ops.attention_x3_supported(sq, skv, 64) == ok, (sq, skv)
        assert ops.attention_x3_blocked(sq, skv) == blocked, (sq, skv)
    assert not ops.attention_x3_supported(1024, 1024, 48)
    # a synthetic trace: steps start with mask_sample_kernel on stream 1 every 10 ms; stream 2 busy the first half of each step
    rows = ["Kind,Start_Timestamp,End_Timestamp,Kernel_Name,Queue_Id,Stream_Id"]
    for st in range(4):
        t0 = st * 10_000_000
        rows.append(f'KERNEL_DISPATCH,{t0},{t0 + 50_000},"mask_sample_kernel(long)",1,1')
        rows.append(f'KERNEL_DISPATCH,{t0 + 100_000},{t0 + 9_900_000},"void g256p::kernel<unsigned short, 0, 0>(g256p::PArgs)",1,1')
        rows.append(f'KERNEL_DISPATCH,{t0},{t0 + 5_000_000},"void cslab::conv_slab_kernel<true>(cdma::Params)",2,2')
    trace = tmp_path / "t_kernel_trace.csv"
    trace.write_text("\n".join(rows) + "\n")
    spec = importlib.util.spec_from_file_location("step_timeline", os.path.join(ROOT, "scripts", "step_timeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old_argv, old_out = sys.argv, sys.stdout
    sys.argv, sys.stdout = ["step_timeline.py", str(trace), "--dump", str(tmp_path / "step.csv")], io.StringIO()
    try:
        mod.main()
        out = sys.stdout.getvalue()
    finally:
        sys.argv, sys.stdout = old_argv, old_out
    assert "step of 10.00 ms" in out and "gemmNN" in out and "conv" in out
    lines = [l for l in out.splitlines() if " ms | " in l]
    assert len(lines) >= 10 and lines[2].split("|")[2].strip().startswith("1.00") and lines[7].split("|")[2].strip().startswith("0.00")
    assert (tmp_path / "step.csv").read_text().count("\n") >= 3
