"""GPU: every HIP kernel, launched through the C ABI (hp_run_op), against the numpy op semantics
(oracle/interp.py) on identical memory images.  Shapes are ragged on purpose (odd lengths,
partial tiles, rows that straddle samples)."""
import numpy as np
import pytest
import torch

from hippie_amd import program as P
from hippie_amd.program import Ref, TapMap
from oracle import interp

pytestmark = pytest.mark.gpu


class Img:
    """One WS arena image shared by the GPU run and the interpreter run."""

    def __init__(self, seed=0):
        self.chunks = []
        self.off = 0
        self.rng = np.random.default_rng(seed)

    def _put(self, arr):
        pad = (-self.off) % 256
        self.off += pad
        self.chunks.append((pad, arr))
        r = Ref(P.WS, self.off)
        self.off += arr.nbytes
        return r

    def f32(self, n, scale=1.0, zero=False):
        a = np.zeros(n, np.float32) if zero else (self.rng.standard_normal(n) * scale).astype(np.float32)
        return self._put(a)

    def f64(self, n):
        return self._put(np.zeros(n, np.float64))

    def i64(self, vals):
        return self._put(np.asarray(vals, dtype=np.int64))

    def image(self):
        out = np.zeros(self.off + 256, np.uint8)
        o = 0
        for pad, arr in self.chunks:
            o += pad
            out[o: o + arr.nbytes] = arr.view(np.uint8).reshape(-1)
            o += arr.nbytes
        return out


def rec_of(op, flags=0, i=(), f=(), buf=()):
    ol = P.OpList()
    ol.add(op, flags, i, f, buf)
    return ol.array()


def run_both(img, recs):
    image = img.image()
    A = interp.Arenas([image.size, 4, 4, 4, 4, 4])
    A.mem[0][:] = image
    interp.run(recs, A)
    dev = torch.from_numpy(image.copy()).cuda()
    bases = [dev.data_ptr()] + [0] * 5
    for r in recs:
        P.run_single_op(r, bases, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dev.cpu().numpy(), A.mem[0]


def view(mem, ref, dtype, n):
    return mem[ref.offset: ref.offset + np.dtype(dtype).itemsize * n].view(dtype)


def check(gpu, cpu, ref, n, dtype=np.float32, rel=2e-5, what=""):
    g, c = view(gpu, ref, dtype, n).astype(np.float64), view(cpu, ref, dtype, n).astype(np.float64)
    scale = max(np.abs(c).max(), 1e-20)
    err = np.abs(g - c).max() / scale
    assert np.isfinite(g).all() and err <= rel, f"{what}: rel err {err:.3e} (scale {scale:.3e})"


R = P.stat_repl       # replicas of a per-channel statistics slot of C channels


def check_stats(gpu, cpu, ref, C, rel=1e-5, what=""):
    """replicated fp64 statistics slot double[R(C)][2][C]: only the sum over replicas is defined"""
    g = view(gpu, ref, np.float64, R(C) * 2 * C).reshape(R(C), 2 * C).sum(0)
    c = view(cpu, ref, np.float64, R(C) * 2 * C).reshape(R(C), 2 * C).sum(0)
    scale = max(np.abs(c).max(), 1e-20)
    err = np.abs(g - c).max() / scale
    assert np.isfinite(g).all() and err <= rel, f"{what}: rel err {err:.3e}"


B = 3
CONV_CASES = {
    # name: (tapmap builder, w_kn, bias, rows_in)
    "fwd_s1": lambda: (TapMap(B * 7, 128, 64, 7, 7, 7, 1, 0, [(t - 1, t) for t in range(3)]), False, False),
    "fwd_s2_odd": lambda: (TapMap(B * 13, 128, 64, 13, 25, 25, 2, 0, [(t - 1, t) for t in range(3)]), False, False),
    "fwd_1x1_s2": lambda: (TapMap(B * 13, 128, 64, 13, 25, 25, 2, 0, [(0, 0)]), False, False),
    "fwd_up_bias": lambda: (TapMap(B * 16, 64, 128, 16, 8, 16, 1, 1, [(t - 1, t) for t in range(3)]), False, True),
    "dgrad_s1": lambda: (TapMap(B * 25, 64, 64, 25, 25, 25, 1, 0, [(1 - t, t) for t in range(3)]), True, False),
    # stride-2 input-gradient by output parity, two sources (conv1 + 1x1 shortcut): planner.map_dgrad_s2_phases
    "dgrad_s2_even": lambda: (TapMap(B * 13, 64, 128, 13, 13, 13, 1, 0, [(0, 1, 0), (0, 0, 1)], out_Lfull=25, out_a=2, out_o=0), True, False),
    "dgrad_s2_odd": lambda: (TapMap(B * 12, 64, 128, 12, 13, 13, 1, 0, [(1, 0, 0), (0, 2, 0)], out_Lfull=25, out_a=2, out_o=1), True, False),
    "dgrad_up": lambda: (TapMap(B * 8, 128, 64, 8, 16, 16, 2, 0, [(e - t + 1, t) for e in (0, 1) for t in range(3)]), True, False),
    "big_k": lambda: (TapMap(37 * 4, 512, 512, 4, 4, 4, 1, 0, [(t - 1, t) for t in range(3)]), False, False),
    "big_k_kn": lambda: (TapMap(37 * 4, 512, 512, 4, 4, 4, 1, 0, [(1 - t, t) for t in range(3)]), True, False),
}


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv_taps(name):
    tm, w_kn, bias = CONV_CASES[name]()
    img = Img(1)
    nb = tm.M // tm.Lout
    a = img.f32(nb * tm.Lin * tm.K)
    nslab = max(t[1] for t in tm.taps) + 1
    w = img.f32(nslab * tm.N * tm.K, scale=0.1)
    two = any(len(t) > 2 and t[2] for t in tm.taps)
    a2 = img.f32(nb * tm.Lin * tm.K) if two else None
    w2 = img.f32(nslab * tm.N * tm.K, scale=0.1) if two else None
    out = img.f32(tm.out_rows * tm.N, scale=3.0)       # pre-filled: rows the op does not own must survive
    bv = img.f32(tm.N) if bias else None
    st = img.f64(R(tm.N) * 2 * tm.N)
    flags = (P.CONV_W_KN if w_kn else 0) | (P.CONV_BIAS if bias else 0) | P.CONV_STATS
    recs = rec_of(P.CONV_TAPS, flags, tm.conv_ints(), (), [a, w, out, bv, st, None, None, None, None, None, a2, w2])
    gpu, cpu = run_both(img, recs)
    check(gpu, cpu, out, tm.out_rows * tm.N, what=name + " out")
    check_stats(gpu, cpu, st, tm.N, what=name + " stats")


def test_conv_stride2_phases_equal_masked_reference():
    """The even-row and odd-row ops of planner.map_dgrad_s2_phases together are the input-gradient of a k=3 stride-2
    pad-1 conv plus that of a 1x1 stride-2 shortcut, written into ONE tensor — against a direct numpy transpose-conv."""
    from hippie_amd import planner
    Bn, Lx, cin, cout = 3, 25, 64, 128
    Ly = (Lx - 1) // 2 + 1
    low = planner.Lowering(planner.ModelCfg(), Bn)
    tms = low.map_dgrad_s2_phases(Lx, Ly, cin, cout)
    img = Img(31)
    dy, dys = img.f32(Bn * Ly * cout), img.f32(Bn * Ly * cout)
    w3, w1 = img.f32(3 * cout * cin, scale=0.1), img.f32(cout * cin, scale=0.1)      # [tap][cout][cin] read as [K][N]; [cout][cin]
    dx = img.f32(Bn * Lx * cin, zero=True)
    ol = P.OpList()
    for tm in tms:
        ol.add(P.CONV_TAPS, P.CONV_W_KN, tm.conv_ints(), (), [dy, w3, dx, None, None, None, None, None, None, None, dys, w1])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, dx, Bn * Lx * cin, what="phases vs interpreter")
    DY = view(cpu, dy, np.float32, Bn * Ly * cout).reshape(Bn, Ly, cout).astype(np.float64)
    DYS = view(cpu, dys, np.float32, Bn * Ly * cout).reshape(Bn, Ly, cout).astype(np.float64)
    W3 = view(cpu, w3, np.float32, 3 * cout * cin).reshape(3, cout, cin).astype(np.float64)
    W1 = view(cpu, w1, np.float32, cout * cin).reshape(cout, cin).astype(np.float64)
    want = np.zeros((Bn, Lx, cin))
    for l in range(Ly):
        for t in range(3):
            p_ = 2 * l + t - 1
            if 0 <= p_ < Lx:
                want[:, p_] += DY[:, l] @ W3[t]
        want[:, 2 * l] += DYS[:, l] @ W1
    got = view(gpu, dx, np.float32, Bn * Lx * cin).reshape(Bn, Lx, cin)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


def _bn_setup(img, M, C):
    """raw tensor + the statistics slot its producer would have accumulated + BatchNorm parameters / buffers"""
    raw = img.f32(M * C)
    gamma, beta, rm = img.f32(C), img.f32(C), img.f32(C, 0.1)
    rv = img._put(np.abs(img.rng.standard_normal(C)).astype(np.float32) + 0.5)
    st = img.f64(R(C) * 2 * C)
    return raw, gamma, beta, rm, rv, st


def _chunk_array(img, ref):
    o = 0
    for pad, arr in img.chunks:
        o += pad
        if o == ref.offset:
            return arr
        o += arr.nbytes
    raise KeyError


@pytest.mark.parametrize("w_kn,up", [(False, False), (False, True), (True, False)])
def test_conv_in_bn_equals_bn_apply_then_conv(w_kn, up):
    """HP_CONV_IN_BN: conv over lrelu(bn(raw)) evaluated in the operand loader == HP_OP_BN_APPLY then the plain conv,
    including the BatchNorm's side effects (saved mean / invstd, running statistics) and the stored (scale, shift)."""
    Bn, Lin, K, N = 5, 9, 128, 64
    tm = (TapMap(Bn * 2 * Lin, N, K, 2 * Lin, Lin, 2 * Lin, 1, 1, [(t - 1, t) for t in range(3)]) if up else
          TapMap(Bn * Lin, N, K, Lin, Lin, Lin, 1, 0, [(t - 1, t) for t in range(3)]))
    M = Bn * Lin
    img = Img(41)
    raw, gamma, beta, rm, rv, st = _bn_setup(img, M, K)
    rm2 = img._put(_chunk_array(img, rm).copy())
    rv2 = img._put(_chunk_array(img, rv).copy())
    r = _chunk_array(img, raw).reshape(M, K).astype(np.float64)
    _chunk_array(img, st)[:K] = r.sum(0)
    _chunk_array(img, st)[K: 2 * K] = (r * r).sum(0)
    w = img.f32(3 * N * K, scale=0.1)
    act = img.f32(M * K, zero=True)
    save_a, save_b, coef = img.f32(2 * K, zero=True), img.f32(2 * K, zero=True), img.f32(2 * K, zero=True)
    out_a, out_b = img.f32(tm.M * N, zero=True), img.f32(tm.M * N, zero=True)
    sta, stb = img.f64(R(N) * 2 * N), img.f64(R(N) * 2 * N)
    fl = (P.CONV_W_KN if w_kn else 0) | P.CONV_STATS
    ol = P.OpList()
    ol.add(P.BN_APPLY, 0, [M, K, 0, 1, 1], [0.01, 1e-5, 0.1], [raw, act, st, gamma, beta, rm, rv, save_a])
    ol.add(P.CONV_TAPS, fl, tm.conv_ints(), (), [act, w, out_a, None, sta])
    ol.add(P.CONV_TAPS, fl | P.CONV_IN_BN, tm.conv_ints() + [M, 0], [0, 0, 0.01, 1e-5, 0.1],
           [raw, w, out_b, None, stb, gamma, beta, rm2, rv2, None, None, None, st, save_b, coef])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, out_b, tm.M * N, what="in-bn conv vs interpreter")
    check_stats(gpu, cpu, stb, N, what="in-bn conv stats")
    for ref, n in ((save_b, 2 * K), (coef, 2 * K), (rm2, K), (rv2, K)):
        check(gpu, cpu, ref, n, rel=1e-6, what="in-bn side effects")
    g = lambda ref, n: view(gpu, ref, np.float32, n)
    # on the GPU the fused op must reproduce the two-op sequence BIT FOR BIT (same coefficient code, same fma, same K order)
    np.testing.assert_array_equal(g(out_a, tm.M * N), g(out_b, tm.M * N))
    np.testing.assert_array_equal(g(save_a, 2 * K), g(save_b, 2 * K))
    np.testing.assert_array_equal(g(rm, K), g(rm2, K))
    np.testing.assert_array_equal(g(rv, K), g(rv2, K))


@pytest.mark.parametrize("variant", ["act_g2", "coef", "second", "phases"])
def test_conv_epilogue_bn_reduce_equals_conv_then_reduce(variant):
    """HP_CONV_EPI_BNRED == plain input-gradient conv followed by HP_OP_BN_BWD_REDUCE."""
    Bn, L, K, N = 4, 13, 128, 64
    M = Bn * L
    img = Img(43)
    dy, w = img.f32(M * K), img.f32(3 * N * K, scale=0.1)
    tm = TapMap(M, N, K, L, L, L, 1, 0, [(1 - t, t) for t in range(3)])
    raw, raw2, act, g2 = img.f32(M * N), img.f32(M * N), img.f32(M * N), img.f32(M * N)
    save, save2, coef = img.f32(2 * N), img.f32(2 * N), img.f32(2 * N)
    tmp = img.f32(M * N, zero=True)
    ga, gb = img.f32(M * N, zero=True), img.f32(M * N, zero=True)
    bsa, bsb, bs2a, bs2b = (img.f64(R(N) * 2 * N) for _ in range(4))
    has_g2 = variant == "act_g2"
    use_coef = variant == "coef"
    second = variant == "second"
    ol = P.OpList()
    tms = [tm]
    if variant == "phases":         # two ops with scattered output rows accumulating into the same statistics slot
        tms = [TapMap(Bn * 7, N, K, 7, L, L, 1, 0, [(0, 1)], out_Lfull=L, out_a=2, out_o=0),
               TapMap(Bn * 6, N, K, 6, L, L, 1, 0, [(1, 0), (0, 2)], out_Lfull=L, out_a=2, out_o=1)]
    for t_ in tms:
        ol.add(P.CONV_TAPS, P.CONV_W_KN, t_.conv_ints(), (), [dy, w, tmp])
    ol.add(P.BN_BWD_REDUCE, 0, [M, N, 1 if has_g2 else 0, 1 if second else 0], [0.01],
           [tmp, g2 if has_g2 else None, None if use_coef else act, ga, raw, save, bsa,
            raw2 if second else None, save2 if second else None, bs2a if second else None, coef if use_coef else None])
    for t_ in tms:
        ol.add(P.CONV_TAPS, P.CONV_W_KN | P.CONV_EPI_BNRED, t_.conv_ints(), [0, 0, 0, 0, 0, 0.01],
               [dy, w, gb] + [None] * 12 + [g2 if has_g2 else None, None if use_coef else act, raw, save, coef if use_coef else None, bsb,
                                            raw2 if second else None, save2 if second else None, bs2b if second else None])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, gb, M * N, what="fused g vs interpreter")
    check_stats(gpu, cpu, bsb, N, what="fused bs vs interpreter")
    g = lambda ref, n, dt=np.float32: view(gpu, ref, dt, n)
    np.testing.assert_array_equal(g(ga, M * N), g(gb, M * N))           # same expressions on the same accumulators
    sa = g(bsa, R(N) * 2 * N, np.float64).reshape(R(N), -1).sum(0)
    sb = g(bsb, R(N) * 2 * N, np.float64).reshape(R(N), -1).sum(0)
    np.testing.assert_allclose(sb, sa, rtol=1e-12, atol=1e-9)
    if second:
        s2a = g(bs2a, R(N) * 2 * N, np.float64).reshape(R(N), -1).sum(0)
        s2b = g(bs2b, R(N) * 2 * N, np.float64).reshape(R(N), -1).sum(0)
        np.testing.assert_allclose(s2b, s2a, rtol=1e-12, atol=1e-9)


def test_conv_mfma_layout_identity():
    """A = I (per tap 1 of a 1-tap map) with an ASYMMETRIC weight matrix: catches transposed C/D or
    swapped operand layouts that random data with loose tolerance could hide."""
    K = N = 64
    tm = TapMap(64, N, K, 64, 64, 64, 1, 0, [(0, 0)])
    for w_kn in (False, True):
        img = Img(2)
        a = img._put(np.eye(64, dtype=np.float32).reshape(-1))
        wmat = (np.arange(N)[:, None] * 1000 + np.arange(K)[None, :]).astype(np.float32)   # W[n][k] = 1000n + k
        w = img._put((wmat.T.copy() if w_kn else wmat).reshape(-1))
        out = img.f32(64 * N, zero=True)
        recs = rec_of(P.CONV_TAPS, P.CONV_W_KN if w_kn else 0, tm.conv_ints(), (), [a, w, out, None, None])
        gpu, _ = run_both(img, recs)
        got = view(gpu, out, np.float32, 64 * N).reshape(64, N)
        np.testing.assert_array_equal(got, wmat.T)     # out[m][n] = sum_k I[m][k] W[n][k] = W[n][m]


WGRAD_CASES = {
    "s1": lambda: TapMap(B * 25, 64, 64, 25, 25, 25, 1, 0, [(t - 1, t) for t in range(3)]),
    "s2": lambda: TapMap(B * 13, 128, 64, 13, 25, 25, 2, 0, [(t - 1, t) for t in range(3)]),
    "1x1": lambda: TapMap(B * 13, 128, 64, 13, 25, 25, 2, 0, [(0, 0)]),
    "up": lambda: TapMap(B * 16, 64, 128, 16, 8, 16, 1, 1, [(t - 1, t) for t in range(3)]),
    "big": lambda: TapMap(50 * 7, 256, 256, 7, 7, 7, 1, 0, [(t - 1, t) for t in range(3)]),
}


@pytest.mark.parametrize("name", list(WGRAD_CASES))
@pytest.mark.parametrize("nsplit", [1, 4])
def test_wgrad_taps_atomic(name, nsplit):
    """flags=1: splits accumulate straight into the (pre-zeroed, here pre-filled) gradient tensor."""
    tm = WGRAD_CASES[name]()
    img = Img(13)
    nb = tm.M // tm.Lout
    dy = img.f32(tm.M * tm.N)
    x = img.f32(nb * tm.Lin * tm.K)
    numel = len(tm.taps) * tm.N * tm.K
    rps = -(-(-(-tm.M // nsplit)) // 32) * 32
    ns = -(-tm.M // rps)
    grad = img.f32(numel, scale=0.5)          # accumulates on top of what is there
    recs = rec_of(P.WGRAD_TAPS, 1, tm.ints() + [ns, rps, numel], (), [dy, x, grad])
    gpu, cpu = run_both(img, recs)
    check(gpu, cpu, grad, numel, rel=3e-5, what=f"wgrad atomic {name}")


@pytest.mark.parametrize("name", ["s1", "up"])
def test_wgrad_in_bn_equals_wgrad_of_the_stored_activation(name):
    """WGRAD_TAPS with HP_CONV_IN_BN: X is the raw BatchNorm input, the operand lrelu(fma(x, scale, shift)) is
    re-evaluated in the loader — bit-identical to the gradient computed from the stored activation."""
    tm = WGRAD_CASES[name]()
    img = Img(17)
    nb = tm.M // tm.Lout
    dy = img.f32(tm.M * tm.N)
    raw = img.f32(nb * tm.Lin * tm.K)
    coef = img.f32(2 * tm.K)
    r = _chunk_array(img, raw).reshape(-1, tm.K)
    cf = _chunk_array(img, coef)
    pre = (r.astype(np.float64) * cf[None, :tm.K].astype(np.float64) + cf[None, tm.K:].astype(np.float64)).astype(np.float32)
    act = img._put(np.where(pre > 0, pre, pre * np.float32(0.01)).astype(np.float32).reshape(-1))
    numel = len(tm.taps) * tm.N * tm.K
    rps = -(-tm.M // 32) * 32
    sa, sb = img.f32(numel, zero=True), img.f32(numel, zero=True)
    ol = P.OpList()
    ol.add(P.WGRAD_TAPS, 0, tm.ints() + [1, rps, numel], [0.01], [dy, act, sa])
    ol.add(P.WGRAD_TAPS, P.CONV_IN_BN, tm.ints() + [1, rps, numel], [0.01], [dy, raw, sb, coef])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, sb, numel, rel=3e-5, what="wgrad in-bn vs interpreter")
    np.testing.assert_array_equal(view(gpu, sa, np.float32, numel), view(gpu, sb, np.float32, numel))


@pytest.mark.parametrize("name", list(WGRAD_CASES))
@pytest.mark.parametrize("nsplit", [1, 3])
def test_wgrad_taps(name, nsplit):
    tm = WGRAD_CASES[name]()
    img = Img(3)
    nb = tm.M // tm.Lout
    dy = img.f32(tm.M * tm.N)
    x = img.f32(nb * tm.Lin * tm.K)
    numel = len(tm.taps) * tm.N * tm.K
    rps = -(-(-(-tm.M // nsplit)) // 32) * 32
    ns = -(-tm.M // rps)
    slab = img.f32(ns * numel, zero=True)
    grad = img.f32(numel, zero=True)
    ol = P.OpList()
    ol.add(P.WGRAD_TAPS, 0, tm.ints() + [ns, rps, numel], (), [dy, x, slab])
    ol.add(P.SLAB_REDUCE, 0, [numel, ns, numel], (), [slab, grad])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, grad, numel, rel=3e-5, what=f"wgrad {name}")


def test_wgrad_mfma_layout_identity():
    """DY = I, X asymmetric: dW[n][k] = sum_m I[m][n] X[m][k] = X[n][k] exactly."""
    tm = TapMap(64, 64, 64, 64, 64, 64, 1, 0, [(0, 0)])
    img = Img(4)
    dy = img._put(np.eye(64, dtype=np.float32).reshape(-1))
    xm = (np.arange(64)[:, None] * 1000 + np.arange(64)[None, :]).astype(np.float32)
    x = img._put(xm.reshape(-1))
    slab = img.f32(64 * 64, zero=True)
    recs = rec_of(P.WGRAD_TAPS, 0, tm.ints() + [1, 64, 64 * 64], (), [dy, x, slab])
    gpu, _ = run_both(img, recs)
    np.testing.assert_array_equal(view(gpu, slab, np.float32, 64 * 64).reshape(64, 64), xm)


# (the last three: workgroups several row batches tall in the reduce — few statistics replicas at C = 512 / 256 — and the
#  scalar form with more than one row per thread)
@pytest.mark.parametrize("C,M", [(64, 300), (512, 37), (20, 77), (5, 33), (512, 1500), (256, 5000), (10, 3000)])
@pytest.mark.parametrize("res_mode", [0, 1, 2])
def test_bn_apply_and_backward(C, M, res_mode):
    img = Img(5)
    raw, raw2 = img.f32(M * C), img.f32(M * C)
    res = img.f32(M * C)
    out = img.f32(M * C, zero=True)
    gamma, beta, gamma2, beta2 = img.f32(C), img.f32(C), img.f32(C), img.f32(C)
    rm, rm2 = img.f32(C, 0.1), img.f32(C, 0.1)
    rv, rv2 = img._put(np.abs(img.rng.standard_normal(C)).astype(np.float32) + 0.5), img._put(np.abs(img.rng.standard_normal(C)).astype(np.float32) + 0.5)
    save, save2 = img.f32(2 * C, zero=True), img.f32(2 * C, zero=True)
    st, st2 = img.f64(R(C) * 2 * C), img.f64(R(C) * 2 * C)
    # statistics as the producing conv would have accumulated them (patched into the fp64 chunks)
    def chunk_array(ref):
        o = 0
        for pad, arr in img.chunks:
            o += pad
            if o == ref.offset:
                return arr
            o += arr.nbytes
        raise KeyError
    for s_ref, r_ref in ((st, raw), (st2, raw2)):
        r = chunk_array(r_ref).reshape(M, C).astype(np.float64)
        chunk_array(s_ref)[:C] = r.sum(0)
        chunk_array(s_ref)[C: 2 * C] = (r * r).sum(0)
    g1, g2 = img.f32(M * C), img.f32(M * C)
    gout, dr, dr2 = img.f32(M * C, zero=True), img.f32(M * C, zero=True), img.f32(M * C, zero=True)
    bs, bs2 = img.f64(R(C) * 2 * C), img.f64(R(C) * 2 * C)
    dgam, dbet, dgam2, dbet2 = (img.f32(C, zero=True) for _ in range(4))
    ol = P.OpList()
    bufs = [raw, out, st, gamma, beta, rm, rv, save]
    if res_mode == 1:
        bufs += [res]
    elif res_mode == 2:
        bufs += [raw2, st2, gamma2, beta2, rm2, rv2, save2]
    ol.add(P.BN_APPLY, 0, [M, C, res_mode, 1, 1], [0.01, 1e-5, 0.1], bufs)
    second = res_mode == 2
    ol.add(P.BN_BWD_REDUCE, 0, [M, C, 1, 1 if second else 0], [0.01],
           [g1, g2, out, gout, raw, save, bs] + ([raw2, save2, bs2] if second else []))
    ol.add(P.BN_BWD_APPLY, 0, [M, C], (), [gout, raw, save, bs, gamma, dr, dgam, dbet])
    if second:
        ol.add(P.BN_BWD_APPLY, 0, [M, C], (), [gout, raw2, save2, bs2, gamma2, dr2, dgam2, dbet2])
    gpu, cpu = run_both(img, ol.array())
    for ref, n, nm in ((out, M * C, "out"), (save, 2 * C, "save"), (rm, C, "rmean"), (rv, C, "rvar"), (gout, M * C, "g"),
                       (dr, M * C, "dr"), (dgam, C, "dgamma"), (dbet, C, "dbeta")):
        check(gpu, cpu, ref, n, rel=3e-5, what=f"bn {nm}")
    check_stats(gpu, cpu, bs, C, what="bn bs")
    if res_mode == 2:
        check(gpu, cpu, dr2, M * C, rel=3e-5, what="bn dr2")
        check(gpu, cpu, rv2, C, rel=3e-5, what="bn rvar2")


def test_bn_eval_mode():
    C, M = 64, 100
    img = Img(6)
    raw, out = img.f32(M * C), img.f32(M * C, zero=True)
    gamma, beta, rm = img.f32(C), img.f32(C), img.f32(C, 0.1)
    rv = img._put(np.abs(img.rng.standard_normal(C)).astype(np.float32) + 0.5)
    save = img.f32(2 * C, zero=True)
    recs = rec_of(P.BN_APPLY, 0, [M, C, 0, 0, 1], [0.2, 1e-5, 0.1], [raw, out, None, gamma, beta, rm, rv, save])
    gpu, cpu = run_both(img, recs)
    check(gpu, cpu, out, M * C, what="bn eval out")
    check(gpu, cpu, rm, C, rel=0, what="running stats untouched in eval")


@pytest.mark.parametrize("Bn,Lin", [(5, 50), (3, 33), (4, 100)])
def test_stem(Bn, Lin):
    Lout = (Lin - 1) // 2 + 1
    img = Img(7)
    x, w = img.f32(Bn * Lin), img.f32(64 * 3)
    out, st = img.f32(Bn * Lout * 64, zero=True), img.f64(R(64) * 128)
    dr, dw = img.f32(Bn * Lout * 64), img.f32(192, zero=True)
    ol = P.OpList()
    ol.add(P.STEM_FWD, 0, [Bn, Lin, Lout, 64], (), [x, w, out, st])
    ol.add(P.STEM_WGRAD, 0, [Bn, Lin, Lout, 64], (), [dr, x, dw])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, out, Bn * Lout * 64, what="stem out")
    check_stats(gpu, cpu, st, 64, what="stem stats")
    check(gpu, cpu, dw, 192, rel=3e-5, what="stem dW")


def test_tail_pool_repeat():
    Bn, Lh, C = 5, 32, 64
    img = Img(8)
    act, w, bias = img.f32(Bn * Lh * C), img.f32(C * 3), img.f32(1)
    t = img.f32(Bn * 2 * Lh, zero=True)
    dt = img.f32(Bn * 2 * Lh)
    dact = img.f32(Bn * Lh * C, zero=True)
    dw, db = img.f32(C * 3, zero=True), img.f32(1, zero=True)
    pooled, gp = img.f32(Bn * C, zero=True), img.f32(Bn * Lh * C, zero=True)
    rep, drep = img.f32(Bn * 4 * C, zero=True), img.f32(Bn * C, zero=True)
    g1, g2 = img.f32(Bn * 4 * C), img.f32(Bn * 4 * C)
    ol = P.OpList()
    ol.add(P.TAIL_FWD, 0, [Bn, Lh, C], (), [act, w, bias, t])
    ol.add(P.TAIL_BWD_X, 0, [Bn, Lh, C], (), [dt, w, dact])
    ol.add(P.TAIL_BWD_W, 0, [Bn, Lh, C], (), [dt, act, dw, db])
    ol.add(P.POOL_FWD, 0, [Bn, Lh, C], (), [act, pooled])
    ol.add(P.POOL_BWD, 0, [Bn, Lh, C], (), [pooled, gp])
    ol.add(P.REPEAT_FWD, 0, [Bn, 4, C], (), [pooled, rep])
    ol.add(P.REPEAT_BWD, 0, [Bn, 4, C, 1], (), [g1, g2, drep])
    gpu, cpu = run_both(img, ol.array())
    for ref, n, nm in ((t, Bn * 2 * Lh, "tail"), (dact, Bn * Lh * C, "tail dX"), (dw, C * 3, "tail dW"), (db, 1, "tail db"),
                       (pooled, Bn * C, "pool"), (gp, Bn * Lh * C, "pool bwd"), (rep, Bn * 4 * C, "repeat"), (drep, Bn * C, "repeat bwd")):
        check(gpu, cpu, ref, n, rel=3e-5, what=nm)


@pytest.mark.parametrize("M,N,K", [(33, 20, 30), (17, 20, 512), (9, 512, 20), (1100, 10, 10), (7, 5, 10)])
def test_linear_family(M, N, K):
    img = Img(9)
    ldx, ldy = K + 3, N + 2
    x, w, bv = img.f32(M * ldx), img.f32(N * K, 0.2), img.f32(N)
    y = img.f32(M * ldy, zero=True)
    st = img.f64(R(N) * 2 * N)
    dy = img.f32(M * ldy)
    dx = img.f32(M * ldx, zero=True)
    dw, db = img.f32(N * K, zero=True), img.f32(N, zero=True)
    ol = P.OpList()
    ol.add(P.LINEAR_FWD, 0, [M, N, K, ldx, ldy, 1, 1], [0.2], [x, w, bv, y, st])
    ol.add(P.LINEAR_BWD_X, 0, [M, N, K, ldy, ldx, 1, ldx, 0], [0.2], [dy, w, dx, x])
    ol.add(P.LINEAR_BWD_X, 0, [M, N, K, ldy, ldx, 0, 0, 1], [0.2], [dy, w, dx, None])
    ol.add(P.LINEAR_BWD_W, 0, [M, N, K, ldy, ldx], (), [dy, x, dw, db])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, y, M * ldy, rel=3e-5, what="linear y")
    check_stats(gpu, cpu, st, N, what="linear stats")
    check(gpu, cpu, dx, M * ldx, rel=3e-5, what="linear dx")
    check(gpu, cpu, dw, N * K, rel=3e-5, what="linear dw")
    check(gpu, cpu, db, N, rel=3e-5, what="linear db")


# (M >= 1024: the three ops run on the matrix cores, csrc/linear_mfma.h.)  The widths are the heads' and the backbones' Linear layers at
# BASELINE configs[2] (z = 32) and [4] (z = 64): 2z + 10 = 74 / 138, 4z + 10 = 266, z + 10 = 42 / 74, 512 <-> 2z, 64 -> output_size;
# pads (+3, +2) give rows that are only 4-byte aligned, (+2, +2) 8-byte, (0, 0) the dense 16-byte case (widths permitting).
@pytest.mark.parametrize("M,N,K,padx,pady", [
    (4096, 64, 512, 0, 0), (4096, 512, 64, 0, 0), (8192, 128, 266, 0, 0), (8192, 128, 138, 3, 2), (4096, 64, 74, 2, 2),
    (8192, 128, 128, 0, 0), (4096, 256, 64, 0, 0), (4096, 32, 64, 3, 2), (1100, 10, 10, 3, 2), (1027, 70, 33, 1, 1), (2048, 20, 15, 0, 0),
])
def test_linear_family_matrix_core_path(M, N, K, padx, pady):
    img = Img(19)
    ldx, ldy = K + padx, N + pady
    x, w, bv = img.f32(M * ldx), img.f32(N * K, 0.2), img.f32(N)
    y = img.f32(M * ldy, zero=True)
    st = img.f64(R(N) * 2 * N)
    dy = img.f32(M * ldy)
    dx = img.f32(M * ldx, zero=True)
    dw, db = img.f32(N * K, zero=True), img.f32(N, zero=True)
    dw1, db1 = img.f32(N * K, zero=True), img.f32(N, zero=True)
    ol = P.OpList()
    ol.add(P.LINEAR_FWD, 0, [M, N, K, ldx, ldy, 1, 1], [0.2], [x, w, bv, y, st])
    ol.add(P.LINEAR_BWD_X, 0, [M, N, K, ldy, ldx, 1, ldx, 0], [0.2], [dy, w, dx, x])
    ol.add(P.LINEAR_BWD_X, 0, [M, N, K, ldy, ldx, 0, 0, 1], [0.2], [dy, w, dx, None])
    ol.add(P.LINEAR_BWD_W, 0, [M, N, K, ldy, ldx], (), [dy, x, dw, db])
    ol.add(P.LINEAR_BWD_W, 1, [M, N, K, ldy, ldx], (), [dy, x, dw1, db1])          # flags & 1: one split per tile, no cross-workgroup atomics
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, y, M * ldy, rel=3e-5, what="linear y")
    check_stats(gpu, cpu, st, N, what="linear stats")
    check(gpu, cpu, dx, M * ldx, rel=3e-5, what="linear dx")
    for a, b in ((dw, db), (dw1, db1)):
        check(gpu, cpu, a, N * K, rel=3e-5, what="linear dw")
        check(gpu, cpu, b, N, rel=3e-5, what="linear db")
    # the padding columns of Y / DX (between N and ldy, K and ldx) belong to other tensors: untouched
    Y = view(gpu, y, np.float32, M * ldy).reshape(M, ldy)
    DX = view(gpu, dx, np.float32, M * ldx).reshape(M, ldx)
    assert not Y[:, N:].any() and not DX[:, K:].any()


def test_linear_matrix_core_path_on_column_windows_and_without_bias():
    """encoder.linear's gradient arrives as a column window of the concatenated head input (planner: `dh = dc0 + 4 * (2z * k)`, leading
    dimension 4z + 10), decoder_fc.0's input-gradient accumulates into one buffer for both towers; no bias / no statistics / no activation."""
    M, N, K, ld = 2048, 128, 512, 266
    img = Img(23)
    cat = img.f32(M * ld)
    win = Ref(P.WS, cat.offset + 4 * 128)                     # columns 128 .. 255 of [M][266]
    x, w = img.f32(M * K), img.f32(N * K, 0.1)
    y = img.f32(M * ld, zero=True)
    ywin = Ref(P.WS, y.offset + 4 * 10)
    dx = img.f32(M * K, zero=True)
    dw, db = img.f32(N * K, zero=True), img.f32(N, zero=True)
    ol = P.OpList()
    ol.add(P.LINEAR_FWD, 0, [M, N, K, K, ld, 0, 0], [0.2], [x, w, None, ywin, None])
    ol.add(P.LINEAR_BWD_X, 0, [M, N, K, ld, K, 0, 0, 0], [0.2], [win, w, dx, None])
    ol.add(P.LINEAR_BWD_W, 0, [M, N, K, ld, K], (), [win, x, dw, None])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, y, M * ld, rel=3e-5, what="windowed y")
    check(gpu, cpu, dx, M * K, rel=3e-5, what="dx from a window")
    check(gpu, cpu, dw, N * K, rel=3e-5, what="dw from a window")
    assert not view(gpu, db, np.float32, N).any()


def test_small_leaf_group_with_matrix_core_members():
    """The small-leaf group at a batch where its Linear members take the matrix-core body (M >= 1024), next to embedding gradients."""
    Bn, H = 2048, 5
    img = Img(67)
    members, outs = [], {}
    for j, (N, K) in enumerate([(128, 266), (64, 128), (128, 74), (256, 64), (128, 512)]):
        dy, x = img.f32(Bn * N), img.f32(Bn * K)
        dw, db = img.f32(N * K, zero=True), img.f32(N, zero=True)
        members.append((P.LINEAR_BWD_W, 0, [Bn, N, K, N, K], (), [dy, x, dw, db]))
        outs[f"dw{j}"], outs[f"db{j}"] = (dw, N * K), (db, N)
    src = img.i64(img.rng.integers(0, 5, Bn))
    dsemb = img.f32(5 * H, zero=True)
    dcat = img.f32(Bn * 74)
    members.append((P.EMB_BWD, 0, [Bn, H, 74, 64, 5], (), [dcat, src, dsemb]))
    outs["dsemb"] = (dsemb, 5 * H)
    single, group = P.OpList(), P.OpList()
    n = len(members)
    for j, (op, fl, i, f, buf) in enumerate(members):
        single.add(op, fl, i, f, buf)
        group.add(op, fl | (P.FLAG_MEMBER if j < n - 1 else ((n - 1) << P.FLAG_GROUP_SHIFT)), i, f, buf)
    gpu_single, cpu = run_both(img, single.array())
    image = img.image()
    dev = torch.from_numpy(image.copy()).cuda()
    prog = P.DeviceProgram(group.array(), [dev.data_ptr()] * 6, [image.size] + [4] * 5)
    seg = prog.capture(0, n)
    prog.replay(seg, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    gpu_group = dev.cpu().numpy()
    for name, (ref, cnt) in outs.items():
        check(gpu_group, cpu, ref, cnt, rel=3e-5, what="group vs interpreter: " + name)
        check(gpu_single, cpu, ref, cnt, rel=3e-5, what="single vs interpreter: " + name)


def test_concat_embedding_reparam_mse_loss():
    Bn, z, H = 37, 10, 5
    img = Img(10)
    h = img.f32(Bn * 2 * z)
    semb, cemb = img.f32(5 * H), img.f32(7 * H)
    src = img.i64(img.rng.integers(1, 5, Bn))
    cls = img.i64(img.rng.integers(0, 7, Bn))
    ld = 2 * z + 2 * H
    c0, c0z = img.f32(Bn * ld, zero=True), img.f32(Bn * ld, zero=True)
    dsemb = img.f32(5 * H, zero=True)
    dcat = img.f32(Bn * ld)
    mulv, eps = img.f32(Bn * 2 * z, 0.5), img.f32(Bn * z)
    zz, dmulv = img.f32(Bn * z, zero=True), img.f32(Bn * 2 * z, zero=True)
    loss = img.f64(4)
    x, rec, drec = img.f32(Bn * 50), img.f32(Bn * 50), img.f32(Bn * 50, zero=True)
    scal = img.f32(4, zero=True)
    ol = P.OpList()
    ol.add(P.CONCAT, 0, [Bn, 3, ld, 0, 0, 2 * z, 2 * z, 1, H, H, 1, H, H, 0, 0, 0, 0, 5, 7], (), [c0, h, None, semb, src, cemb, cls])
    ol.add(P.CONCAT, 0, [Bn, 3, ld, 0, 0, 2 * z, 2 * z, 1, H, H, 2, H, 0, 0, 0, 0, 0, 5, 0], (), [c0z, h, None, semb, src, None, None])
    ol.add(P.EMB_BWD, 0, [Bn, H, ld, 2 * z, 5], (), [dcat, src, dsemb])
    ol.add(P.REPARAM_KL_FWD, 0, [Bn, z], (), [mulv, eps, zz, loss])
    ol.add(P.REPARAM_KL_BWD, 0, [Bn, z, ld], [0.7], [mulv, eps, dcat, dmulv])
    ol.add(P.MSE_FWD_BWD, 0, [Bn * 50, 1], [0.5], [x, rec, drec, loss])
    ol.add(P.LOSS_FINALIZE, 0, [Bn, Bn * 50, 0], [0.7, 1.0, 0.0], [loss, scal])
    gpu, cpu = run_both(img, ol.array())
    for ref, n, nm in ((c0, Bn * ld, "concat"), (c0z, Bn * ld, "concat zeros"), (dsemb, 5 * H, "emb bwd"), (zz, Bn * z, "z"),
                       (dmulv, Bn * 2 * z, "dmulv"), (drec, Bn * 50, "drec"), (scal, 4, "scalars")):
        check(gpu, cpu, ref, n, rel=3e-5, what=nm)
    check(gpu, cpu, loss, 4, np.float64, rel=1e-5, what="loss slots")


def test_out_of_range_labels_are_harmless_on_the_device():
    """nn.Embedding raises IndexError on a bad index (the host side does too: Engine.check_labels); the kernels
    themselves must never fault on one: the gathered row is zeros and its gradient contribution is dropped, and the
    words behind the table stay untouched."""
    Bn, H, rows = 16, 5, 5
    img = Img(12)
    h = img.f32(Bn * 4)
    semb = img.f32(rows * H)
    guard = img.f32(64)                       # sits right behind the table: must not be read into the output ...
    labels = np.array([0, 1, 2, 3, 4, 5, 7, -1, 2**40, -2**40, 4, 3, 2, 1, 0, 100], dtype=np.int64)
    src = img.i64(labels)
    ld = 4 + H
    out = img.f32(Bn * ld, zero=True)
    dcat = img.f32(Bn * ld)
    dsemb = img.f32(rows * H, zero=True)
    guard2 = img.f32(64, zero=True)           # ... and this one sits behind the gradient table: must stay zero
    ol = P.OpList()
    ol.add(P.CONCAT, 0, [Bn, 2, ld, 0, 0, 4, 4, 1, H, H, 0, 0, 0, 0, 0, 0, 0, rows], (), [out, h, None, semb, src])
    ol.add(P.EMB_BWD, 0, [Bn, H, ld, 4, rows], (), [dcat, src, dsemb])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, out, Bn * ld, what="concat with bad labels")
    check(gpu, cpu, dsemb, rows * H, rel=3e-5, what="emb bwd with bad labels")
    o = view(gpu, out, np.float32, Bn * ld).reshape(Bn, ld)
    bad = (labels < 0) | (labels >= rows)
    assert np.all(o[bad, 4:] == 0) and np.all(o[~bad, 4:] != 0)
    assert np.all(view(gpu, guard2, np.float32, 64) == 0)


@pytest.mark.parametrize("n", [1027, 8056614 // 16])
@pytest.mark.parametrize("clip", [0.0, 1.0])
def test_adamw_gradnorm_steps(n, clip):
    img = Img(11)
    p, g = img.f32(n, 0.1), img.f32(n)
    m, v = img.f32(n, zero=True), img.f32(n, zero=True)
    step = img.i64([0])
    norm2 = img.f64(1)
    ol = P.OpList()
    for _ in range(3):
        ol.add(P.ZERO, 0, [8, 0], (), [norm2])
        ol.add(P.GRADNORM, 0, [n], (), [g, norm2])
        ol.add(P.STEP_INC, 0, (), (), [step])
        ol.add(P.ADAMW, 0, [n], [1e-3, 0.9, 0.999, 1e-8, 0.01, clip, 1 - 0.9, 1 - 0.999], [p, g, m, v, step, norm2])
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, norm2, 1, np.float64, rel=1e-9, what="norm2")
    for ref, nm in ((p, "p"), (m, "m"), (v, "v")):
        check(gpu, cpu, ref, n, rel=1e-5, what="adam " + nm)
    assert view(gpu, step, np.int64, 1)[0] == 3
    # and against torch.optim.AdamW itself on the same numbers
    im = img.image()
    tp = torch.tensor(view(im, p, np.float32, n).copy(), requires_grad=True)
    opt = torch.optim.AdamW([tp], lr=1e-3, weight_decay=0.01)
    for _ in range(3):
        tp.grad = torch.tensor(view(im, g, np.float32, n).copy())
        if clip:
            torch.nn.utils.clip_grad_norm_([tp], clip)
        opt.step()
    got = view(gpu, p, np.float32, n)
    np.testing.assert_allclose(got, tp.detach().numpy(), rtol=2e-5, atol=2e-6)


def run_program_both(img, recs):
    """like run_both but through hp_program_create/run (needed for PAIR / WGRAD_GROUP, which reference
    other records by program index)."""
    image = img.image()
    A = interp.Arenas([image.size, 4, 4, 4, 4, 4])
    A.mem[0][:] = image
    interp.run(recs, A)
    dev = torch.from_numpy(image.copy()).cuda()
    dummy = torch.zeros(64, dtype=torch.uint8, device="cuda")
    prog = P.DeviceProgram(recs, [dev.data_ptr()] + [dummy.data_ptr()] * 5, [dev.numel()] + [64] * 5)
    prog.run(0, len(recs), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dev.cpu().numpy(), A.mem[0]


@pytest.mark.parametrize("w_kn", [False, True])
def test_pair_conv_and_bn(w_kn):
    """HP_OP_PAIR: two different-shaped convs / BatchNorm passes in one launch each."""
    img = Img(21)
    tms = [TapMap(B * 25, 64, 64, 25, 25, 25, 1, 0, [((1 - t) if w_kn else (t - 1), t) for t in range(3)]),
           TapMap(5 * 50, 128, 64, 50, 50, 50, 1, 0, [((1 - t) if w_kn else (t - 1), t) for t in range(3)])]
    ol = P.OpList()
    outs = []
    for tm in tms:
        a = img.f32((tm.M // tm.Lout) * tm.Lin * tm.K)
        w = img.f32(3 * tm.N * tm.K, scale=0.1)
        out = img.f32(tm.M * tm.N, zero=True)
        st = img.f64(R(tm.N) * 2 * tm.N)
        ol.add(P.CONV_TAPS, (P.CONV_W_KN if w_kn else 0) | P.CONV_STATS | P.FLAG_MEMBER, tm.conv_ints(), (), [a, w, out, None, st])
        outs.append((tm, out, st))
    ol.add(P.PAIR, 0, [0, 1])
    # BN apply on both conv outputs, paired
    bn = []
    for k, (tm, out, st) in enumerate(outs):
        C, M = tm.N, tm.M
        gamma, beta, rm = img.f32(C), img.f32(C), img.f32(C, 0.1)
        rv = img._put(np.abs(img.rng.standard_normal(C)).astype(np.float32) + 0.5)
        save, y = img.f32(2 * C, zero=True), img.f32(M * C, zero=True)
        ol.add(P.BN_APPLY, P.FLAG_MEMBER, [M, C, 0, 1, 1], [0.01, 1e-5, 0.1], [out, y, st, gamma, beta, rm, rv, save])
        bn.append((M, C, y, save, rm))
    ol.add(P.PAIR, 0, [3, 4])
    gpu, cpu = run_program_both(img, ol.array())
    for tm, out, st in outs:
        check(gpu, cpu, out, tm.M * tm.N, what="pair conv out")
        check_stats(gpu, cpu, st, tm.N, what="pair conv stats")
    for M, C, y, save, rm in bn:
        check(gpu, cpu, y, M * C, rel=3e-5, what="pair bn out")
        check(gpu, cpu, save, 2 * C, rel=3e-5, what="pair bn save")
        check(gpu, cpu, rm, C, rel=3e-5, what="pair bn running mean")


def test_wgrad_group_two_problems():
    img = Img(22)
    ol = P.OpList()
    grads = []
    for tm, nsplit in ((WGRAD_CASES["s1"](), 2), (WGRAD_CASES["up"](), 1), (WGRAD_CASES["big"](), 3)):
        nb = tm.M // tm.Lout
        dy, x = img.f32(tm.M * tm.N), img.f32(nb * tm.Lin * tm.K)
        numel = 3 * tm.N * tm.K
        rps = -(-(-(-tm.M // nsplit)) // 32) * 32
        ns = -(-tm.M // rps)
        g = img.f32(numel, zero=True)
        ol.add(P.WGRAD_TAPS, 1 | P.FLAG_MEMBER, tm.ints() + [ns, rps, numel], (), [dy, x, g])
        grads.append((g, numel))
    ol.add(P.WGRAD_GROUP, 0, [0, 3, 3])
    gpu, cpu = run_program_both(img, ol.array())
    for g, numel in grads:
        check(gpu, cpu, g, numel, rel=3e-5, what="grouped wgrad")


@pytest.mark.parametrize("with_res,act", [(False, True), (True, True), (True, False)])
def test_conv_eval_bn_epilogue_equals_conv_then_bn_apply(with_res, act):
    """CONV_TAPS flag 8 (eval BatchNorm folded into the epilogue) against the interpreter AND, bit for bit,
    against the two-launch form conv -> BN_APPLY(eval) on the GPU."""
    tm = TapMap(B * 13, 128, 64, 13, 25, 25, 2, 0, [(t - 1, t) for t in range(3)])
    M, N, K = tm.M, tm.N, tm.K
    img = Img(5)
    a, w = img.f32(B * 25 * K), img.f32(3 * N * K, 0.05)
    gamma, beta, rmean = img.f32(N, 0.5), img.f32(N, 0.2), img.f32(N, 0.3)
    rvar = img._put((np.abs(img.rng.standard_normal(N)) + 0.5).astype(np.float32))
    res = img.f32(M * N)
    out_f, raw, out_2, save = img.f32(M * N, zero=True), img.f32(M * N, zero=True), img.f32(M * N, zero=True), img.f32(2 * N, zero=True)
    ol = P.OpList()
    ol.add(P.CONV_TAPS, P.CONV_BN_EVAL | (P.CONV_ACT if act else 0), tm.conv_ints(), [1e-5, 0.01],
           [a, w, out_f, None, None, gamma, beta, rmean, rvar, res if with_res else None])
    ol.add(P.CONV_TAPS, 0, tm.conv_ints(), (), [a, w, raw, None, None])
    ol.add(P.BN_APPLY, 0, [M, N, 1 if with_res else 0, 0, 1 if act else 0], [0.01, 1e-5, 0.1],
           [raw, out_2, None, gamma, beta, rmean, rvar, save] + ([res] if with_res else []))
    gpu, cpu = run_both(img, ol.array())
    check(gpu, cpu, out_f, M * N, rel=3e-5, what="fused conv+BN vs interpreter")
    assert np.array_equal(view(gpu, out_f, np.float32, M * N), view(gpu, out_2, np.float32, M * N)), "fused != conv -> BN_APPLY"


def test_parallel_group_of_small_weight_gradients_equals_standalone_launches():
    """Small-leaf group (HP_FLAG_GROUP_SHIFT): independent small leaf ops (Linear weight / bias gradients of different layers, two embedding
    gradients into ONE table) run side by side in one launch — against the interpreter and against one launch per record."""
    Bn, H = 300, 5
    img = Img(61)
    shapes = [(20, 30), (10, 20), (20, 10), (100, 64), (512, 20)]          # (N, K) of the layers
    members, outs = [], {}
    for j, (N, K) in enumerate(shapes):
        dy, x = img.f32(Bn * N), img.f32(Bn * K)
        dw, db = img.f32(N * K, zero=True), img.f32(N, zero=True)
        members.append((P.LINEAR_BWD_W, 0, [Bn, N, K, N, K], (), [dy, x, dw, db]))
        outs[f"dw{j}"] = (dw, N * K)
        outs[f"db{j}"] = (db, N)
    src = img.i64(img.rng.integers(0, 5, Bn))
    dsemb = img.f32(5 * H, zero=True)
    for j in range(2):                                                      # decoder-side and encoder-side contribution
        dcat = img.f32(Bn * 30)
        members.append((P.EMB_BWD, 0, [Bn, H, 30, 20, 5], (), [dcat, src, dsemb]))
    outs["dsemb"] = (dsemb, 5 * H)
    single, group = P.OpList(), P.OpList()
    n = len(members)
    for j, (op, fl, i, f, buf) in enumerate(members):
        single.add(op, fl, i, f, buf)
        group.add(op, fl | (P.FLAG_MEMBER if j < n - 1 else ((n - 1) << P.FLAG_GROUP_SHIFT)), i, f, buf)
    gpu_single, cpu = run_both(img, single.array())
    image = img.image()
    dev = torch.from_numpy(image.copy()).cuda()
    prog = P.DeviceProgram(group.array(), [dev.data_ptr()] * 6, [image.size] + [4] * 5)
    seg = prog.capture(0, n)
    prog.replay(seg, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    gpu_group = dev.cpu().numpy()
    for name, (ref, cnt) in outs.items():
        check(gpu_group, cpu, ref, cnt, rel=3e-5, what="group vs interpreter: " + name)
        a, b = view(gpu_single, ref, np.float32, cnt), view(gpu_group, ref, np.float32, cnt)
        assert np.abs(a - b).max() <= 2e-6 * max(np.abs(a).max(), 1e-30), name       # (fp32 atomics: last bits)
    # a member that is not independent-launchable is refused at program creation
    bad = group.array().copy()
    bad[0]["op"] = P.CONV_TAPS
    with pytest.raises(P.HipEngineError):
        P.DeviceProgram(bad, [dev.data_ptr()] * 6, [image.size] + [4] * 5)


def test_ranges_that_cut_through_a_launch_unit_are_refused():
    """hp_program_run / capture / profile reject a range that holds members without their closing record (they would be
    silently skipped) or a closing record without its members (ops outside the range would run)."""
    Bn = 64
    img = Img(62)
    group = P.OpList()
    for j, (N, K) in enumerate([(20, 30), (10, 20), (20, 10)]):
        dy, x = img.f32(Bn * N), img.f32(Bn * K)
        dw, db = img.f32(N * K, zero=True), img.f32(N, zero=True)
        group.add(P.LINEAR_BWD_W, P.FLAG_MEMBER if j < 2 else (2 << P.FLAG_GROUP_SHIFT), [Bn, N, K, N, K], (), [dy, x, dw, db])
    image = img.image()
    dev = torch.from_numpy(image.copy()).cuda()
    prog = P.DeviceProgram(group.array(), [dev.data_ptr()] * 6, [image.size] + [4] * 5)
    prog.run(0, 3)
    torch.cuda.synchronize()
    for first, count in ((0, 2), (1, 2), (2, 1), (0, 1)):
        with pytest.raises(P.HipEngineError):
            prog.run(first, count)
        with pytest.raises(P.HipEngineError):
            prog.capture(first, count)


@pytest.mark.parametrize("two_tables,world,rank", [(False, 1, 0), (True, 2, 1)])
def test_stage_batch_gathers_rows_and_draws_philox_noise(two_tables, world, rank):
    """HP_OP_STAGE_BATCH against the interpreter: the gathered rows and labels are exact copies, the noise is the interpreter's
    Philox stream (same uint32 words; float32 log / sqrt / sincos may differ in the last bits: 2e-6 absolute), for three
    consecutive cursor values incl. the wrap-around, one and two tables, a rank interleave, an out-of-range index."""
    Bn, L, L2, z, N = 37, 50, 100, 10, 200
    img = Img(81)
    table, table2 = img.f32(N * L), (img.f32(N * L2) if two_tables else None)
    labels = img.i64(img.rng.integers(0, 5, N))
    perm = img.rng.permutation(N)
    perm[5] = N + 3                                     # an index outside the table: reads row 0, never faults
    permr = img.i64(perm)
    cursor, seed = img.i64([0]), img.i64([0x1234ABCD5678])
    x, x2 = img.f32(Bn * L, zero=True), (img.f32(Bn * L2, zero=True) if two_tables else None)
    src, eps = img.i64(np.zeros(Bn)), img.f32(Bn * z, zero=True)
    spe = N // (Bn * world)
    ol = P.OpList()
    ol.add(P.STAGE_BATCH, 0, [Bn, L, L2 if two_tables else 0, z, spe, world, rank, N], (), [table, table2, labels, permr, cursor, x, x2, src, eps, seed])
    ol.add(P.STEP_INC, 0, (), (), [cursor])
    recs = ol.array()
    image = img.image()
    A = interp.Arenas([image.size, 4, 4, 4, 4, 4])
    A.mem[0][:] = image
    dev = torch.from_numpy(image.copy()).cuda()
    bases = [dev.data_ptr()] + [0] * 5
    seen = []
    for step in range(spe + 1):
        interp.run(recs, A)
        for r in recs:
            P.run_single_op(r, bases, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        gpu, cpu = dev.cpu().numpy(), A.mem[0]
        for ref, n, dt in ((x, Bn * L, np.float32), (src, Bn, np.int64), (cursor, 1, np.int64)) + (((x2, Bn * L2, np.float32),) if two_tables else ()):
            np.testing.assert_array_equal(view(gpu, ref, dt, n), view(cpu, ref, dt, n))
        g, c = view(gpu, eps, np.float32, Bn * z), view(cpu, eps, np.float32, Bn * z)
        assert np.abs(g - c).max() <= 2e-6 * max(1.0, np.abs(c).max()), np.abs(g - c).max()
        seen.append(g.copy())
    assert int(view(dev.cpu().numpy(), cursor, np.int64, 1)[0]) == spe + 1
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[0], seen[-1]), "the noise must depend on the cursor, also across the wrap"


def _random_tapmaps(rng, count):
    """Seeded sweep over the tap-map family the planner can emit, with ragged sizes the fixed cases do not have: batch x length
    products that are not multiples of the 64-row tile, column counts that end inside a tile (N % 64 != 0, N % 4 == 0), every K
    from 32 to 512, stride 1 / 2, nearest x2 up-sampling, 1 / 3 / 6 taps, forward and transposed weights."""
    out = []
    while len(out) < count:
        kind = rng.choice(["s1", "s2", "1x1", "up", "dgrad_s1", "dgrad_up"])
        nb = int(rng.integers(1, 6))
        K = int(rng.choice([32, 64, 96, 128, 256, 512]))
        N = int(rng.choice([4, 36, 64, 100, 128, 192, 256]))
        L = int(rng.integers(2, 40))
        if kind == "s1":
            tm, kn = TapMap(nb * L, N, K, L, L, L, 1, 0, [(t - 1, t) for t in range(3)]), False
        elif kind == "s2":
            Lo = (L - 1) // 2 + 1
            tm, kn = TapMap(nb * Lo, N, K, Lo, L, L, 2, 0, [(t - 1, t) for t in range(3)]), False
        elif kind == "1x1":
            Lo = (L - 1) // 2 + 1
            tm, kn = TapMap(nb * Lo, N, K, Lo, L, L, 2, 0, [(0, 0)]), False
        elif kind == "up":
            tm, kn = TapMap(nb * 2 * L, N, K, 2 * L, L, 2 * L, 1, 1, [(t - 1, t) for t in range(3)]), False
        elif kind == "dgrad_s1":
            tm, kn = TapMap(nb * L, N, K, L, L, L, 1, 0, [(1 - t, t) for t in range(3)]), True
        else:
            tm, kn = TapMap(nb * L, N, K, L, 2 * L, 2 * L, 2, 0, [(e - t + 1, t) for e in (0, 1) for t in range(3)]), True
        out.append((kind, tm, kn))
    return out


def test_conv_taps_random_shapes():
    rng = np.random.default_rng(2026)
    for k, (kind, tm, w_kn) in enumerate(_random_tapmaps(rng, 36)):
        img = Img(100 + k)
        nb = tm.M // tm.Lout
        a = img.f32(nb * tm.Lin * tm.K)
        nslab = max(t[1] for t in tm.taps) + 1
        w = img.f32(nslab * tm.N * tm.K, scale=0.1)
        bias = bool(k % 3 == 0)
        out = img.f32(tm.M * tm.N, scale=3.0)
        bv = img.f32(tm.N) if bias else None
        st = img.f64(R(tm.N) * 2 * tm.N)
        flags = (P.CONV_W_KN if w_kn else 0) | (P.CONV_BIAS if bias else 0) | P.CONV_STATS
        recs = rec_of(P.CONV_TAPS, flags, tm.conv_ints(), (), [a, w, out, bv, st])
        gpu, cpu = run_both(img, recs)
        what = f"random conv {k} {kind} M={tm.M} N={tm.N} K={tm.K} L={tm.Lout}"
        check(gpu, cpu, out, tm.M * tm.N, what=what + " out")
        check_stats(gpu, cpu, st, tm.N, what=what + " stats")


def test_wgrad_taps_random_shapes():
    rng = np.random.default_rng(2027)
    for k, (kind, tm, _) in enumerate(_random_tapmaps(rng, 60)):
        if kind.startswith("dgrad") or tm.M < 2:
            continue
        img = Img(300 + k)
        nb = tm.M // tm.Lout
        dy = img.f32(tm.M * tm.N)
        x = img.f32(nb * tm.Lin * tm.K)
        numel = len(tm.taps) * tm.N * tm.K
        nsplit = int(rng.integers(1, 4))
        rps = -(-(-(-tm.M // nsplit)) // 32) * 32
        ns = -(-tm.M // rps)
        grad = img.f32(numel, scale=0.5)
        recs = rec_of(P.WGRAD_TAPS, 1, tm.ints() + [ns, rps, numel], (), [dy, x, grad])
        gpu, cpu = run_both(img, recs)
        check(gpu, cpu, grad, numel, rel=3e-5, what=f"random wgrad {k} {kind} M={tm.M} N={tm.N} K={tm.K} L={tm.Lout} splits={ns}")
