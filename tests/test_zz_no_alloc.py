"""The entry point that allocates nothing (include/vtc_amd.h: vtc_decode_static) against the existing vtc_decode, through
the C ABI, bit for bit: two calls in a row on one GPU, the second with a smaller input (the scratch array keeps the first
call's contents), then the size limit.  Collected last (file name) so that a failure here hides no other test."""

import ctypes as C

import numpy as np
import pytest
import torch

from oracle import vtc_oracle as O
from vtc_amd import lib
from vtc_amd.segma_compat.inference import logit_bounds

pytestmark = pytest.mark.gpu

THR = {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in O.DEFAULT_LABELS}


def _static(eng, L, logits, lo, hi, cap):
    s = torch.full((cap,), -7, dtype=torch.int64, device="cuda")
    e = torch.full((cap,), -7, dtype=torch.int64, device="cuda")
    lab = torch.full((cap,), -7, dtype=torch.int32, device="cuda")
    result = torch.full((1,), -9, dtype=torch.int64, device="cuda")
    summary = torch.zeros(2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rc = L.vtc_decode_static(eng._handle, C.c_void_p(logits.data_ptr()), int(logits.shape[0]), lo.ctypes.data_as(lib._f32p),
                             hi.ctypes.data_as(lib._f32p), C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()),
                             C.c_void_p(lab.data_ptr()), cap, C.c_void_p(result.data_ptr()), C.c_void_p(summary.data_ptr()),
                             eng._stream())
    lib.check(rc, eng._handle)
    torch.cuda.synchronize()
    return s.cpu().numpy(), e.cpu().numpy(), lab.cpu().numpy(), int(result.item()), summary.cpu().numpy()


def test_static_decode_equals_vtc_decode_twice_in_a_row():
    from vtc_amd.engine import VtcEngine
    L = lib.load()
    eng = VtcEngine(O.make_weights(seed=0), n_labels=4, chunk_duration_s=1.0, max_batch=1, device=0)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    rng = np.random.default_rng(12)
    smooth = np.cumsum(rng.standard_normal((200_000, 4)), axis=0).astype(np.float32) * 0.05
    big = torch.from_numpy(np.ascontiguousarray(smooth - smooth.mean(0))).cuda()
    small = torch.from_numpy(rng.standard_normal((3_001, 4)).astype(np.float32)).cuda()
    for logits in (big, small, big[:777].contiguous(), small):          # larger, then smaller, inputs on the same scratch
        ws, we, wl, wmax, wmean = eng.decode(logits, lo, hi)            # the existing entry point (vtc_decode)
        cap = len(ws) + 5
        s, e, lab, n, summary = _static(eng, L, logits, lo, hi, cap)
        assert n == len(ws)
        assert np.array_equal(s[:n], ws) and np.array_equal(e[:n], we) and np.array_equal(lab[:n], wl)
        assert np.all(s[n:] == -7) and np.all(lab[n:] == -7)            # nothing written past the runs
        assert summary[0] == wmax and summary[1] == wmean               # same kernels, same order of summation
        # against the oracle as well (index work: bit exact)
        want = O.create_intervals(O.apply_thresholds(logits.cpu(), THR))
        assert [(int(a), int(b), O.DEFAULT_LABELS[int(c)]) for a, b, c in zip(s[:n], e[:n], lab[:n])] == want
    # The property the entry point exists for: it neither allocates nor synchronises.  (1) device memory in use is the same
    # before and after a call; (2) a call made while its stream is being captured into a HIP graph succeeds (an allocation,
    # a blocking copy or a synchronisation there would fail the capture), and replaying the graph on new logits in the same
    # buffer gives vtc_decode's answer for those logits.
    free0 = torch.cuda.mem_get_info()[0]
    _static(eng, L, small, lo, hi, 4096)
    assert torch.cuda.mem_get_info()[0] == free0
    buf = small.clone()
    cap = 4096
    gs = torch.full((cap,), -7, dtype=torch.int64, device="cuda")
    ge = torch.full((cap,), -7, dtype=torch.int64, device="cuda")
    gl = torch.full((cap,), -7, dtype=torch.int32, device="cuda")
    gres = torch.full((1,), -9, dtype=torch.int64, device="cuda")
    gsum = torch.zeros(2, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = L.vtc_decode_static(eng._handle, C.c_void_p(buf.data_ptr()), int(buf.shape[0]), lo.ctypes.data_as(lib._f32p),
                                 hi.ctypes.data_as(lib._f32p), C.c_void_p(gs.data_ptr()), C.c_void_p(ge.data_ptr()),
                                 C.c_void_p(gl.data_ptr()), cap, C.c_void_p(gres.data_ptr()), C.c_void_p(gsum.data_ptr()),
                                 eng._stream())
        lib.check(rc, eng._handle)
    for trial in range(2):
        fresh = torch.from_numpy(rng.standard_normal((3_001, 4)).astype(np.float32)).cuda()
        buf.copy_(fresh)
        graph.replay()
        torch.cuda.synchronize()
        ws, we, wl, wmax, wmean = eng.decode(fresh, lo, hi)
        n = int(gres.item())
        assert n == len(ws) and np.array_equal(gs.cpu().numpy()[:n], ws) and np.array_equal(ge.cpu().numpy()[:n], we)
        assert np.array_equal(gl.cpu().numpy()[:n], wl) and gsum[0].item() == wmax and gsum[1].item() == wmean
    del graph
    # room for fewer runs than there are: the count is still reported, only `cap` runs are written
    ws, we, wl, _, _ = eng.decode(small, lo, hi)
    s, e, lab, n, _ = _static(eng, L, small, lo, hi, 10)
    assert n == len(ws) > 10 and np.array_equal(s, ws[:10]) and np.array_equal(lab, wl[:10])
    # non-finite logits are marked in the summary (as vtc_decode_files does)
    bad = small.clone()
    bad[100, 1] = float("nan")
    assert np.isnan(_static(eng, L, bad, lo, hi, 4096)[4][0])
    # sizes beyond the stated maximum are rejected with a clear error, nothing is launched
    too_big = torch.zeros((lib.STATIC_DECODE_MAX_FRAMES + 1, 4), device="cuda")
    with pytest.raises(lib.VtcError) as ei:
        _static(eng, L, too_big, lo, hi, 16)
    assert ei.value.status == lib.ERR_CAPACITY and "at most" in str(ei.value)
    at_limit = torch.full((lib.STATIC_DECODE_MAX_FRAMES, 4), -1.0, device="cuda")
    at_limit[::2] = 1.0                                                 # the maximum number of runs per label
    s, e, lab, n, _ = _static(eng, L, at_limit, lo, hi, 4 * (lib.STATIC_DECODE_MAX_FRAMES // 2))
    assert n == 4 * (lib.STATIC_DECODE_MAX_FRAMES // 2)
    assert np.array_equal(s[: n // 4], np.arange(0, lib.STATIC_DECODE_MAX_FRAMES, 2) * 320) and np.all(e - s == 400)
    eng.close()


def test_static_decode_refuses_a_call_that_would_overlap_the_previous_one():
    """All calls on a GPU share one scratch array (include/vtc_amd.h).  A call on another stream while the previous call's
    kernels are pending is refused with VTC_ERR_ARG; the same stream, or the other stream once the work is done, is fine."""
    from vtc_amd.engine import VtcEngine
    L = lib.load()
    eng = VtcEngine(O.make_weights(seed=0), n_labels=4, chunk_duration_s=1.0, max_batch=1, device=0)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    logits = torch.from_numpy(np.random.default_rng(5).standard_normal((3_001, 4)).astype(np.float32)).cuda()
    want = eng.decode(logits, lo, hi)
    cap = 4096
    outs = [(torch.full((cap,), -7, dtype=torch.int64, device="cuda"), torch.full((cap,), -7, dtype=torch.int64, device="cuda"),
             torch.full((cap,), -7, dtype=torch.int32, device="cuda"), torch.full((1,), -9, dtype=torch.int64, device="cuda"),
             torch.zeros(2, dtype=torch.float64, device="cuda")) for _ in range(4)]

    def call(k, stream):
        s, e, lab, res, summ = outs[k]
        return L.vtc_decode_static(eng._handle, C.c_void_p(logits.data_ptr()), int(logits.shape[0]), lo.ctypes.data_as(lib._f32p),
                                   hi.ctypes.data_as(lib._f32p), C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()),
                                   C.c_void_p(lab.data_ptr()), cap, C.c_void_p(res.data_ptr()), C.c_void_p(summ.data_ptr()),
                                   C.c_void_p(stream.cuda_stream))

    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(20_000_000)          # stream a is busy for >= 10 ms: the first call's kernels wait behind it
    assert call(0, a) == lib.OK
    rc = call(1, b)                            # would run beside the first call's kernels on the same scratch
    assert rc == lib.ERR_ARG
    with pytest.raises(lib.VtcError) as ei:
        lib.check(rc, eng._handle)
    assert "must not overlap" in str(ei.value)
    assert call(2, a) == lib.OK                # same stream: ordered behind the first call
    torch.cuda.synchronize()
    assert call(3, b) == lib.OK                # nothing pending any more
    torch.cuda.synchronize()
    assert int(outs[1][3].item()) == -9        # the refused call launched nothing
    for k in (0, 2, 3):
        s, e, lab, res, summ = outs[k]
        n = int(res.item())
        assert n == len(want[0]) and np.array_equal(s.cpu().numpy()[:n], want[0]) and np.array_equal(e.cpu().numpy()[:n], want[1])
        assert np.array_equal(lab.cpu().numpy()[:n], want[2]) and summ[0].item() == want[3] and summ[1].item() == want[4]
    eng.close()
