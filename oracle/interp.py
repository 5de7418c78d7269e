"""numpy interpreter of HpOp programs — TEST INFRASTRUCTURE ONLY (never imported by hippie_amd).

An executable restatement of the op semantics documented in include/hippie_hip.h.
tests/ run the planner's program through it on the CPU and compare with the
torch oracle (oracle/cvae_oracle.py), which is itself pinned to the reference by
tests/golden/.  That validates the host-side lowering (arena layout, tap maps,
backward wiring) without a GPU; the GPU tests then compare each HIP kernel with
these same semantics.
"""
from __future__ import annotations

import numpy as np

NULL = -1
STAT_REPL_MAX = 16
MASK = (1 << 56) - 1
CONV_IN_BN, CONV_EPI_BNRED, CONV_BF16 = 64, 128, 256


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on uint32 numpy arrays (counter words c0..c3, key k0, k1) -> four uint32 arrays."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in (c0, c1, c2, c3)]


def philox_normal(seed, cursor, n, rank=0):
    """HP_OP_STAGE_BATCH's noise: n standard normals for (seed, cursor, data-parallel rank) (float32; Box-Muller on 24-bit uniforms)."""
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    seed, cursor = int(seed) & 0xFFFFFFFFFFFFFFFF, int(cursor) & 0xFFFFFFFFFFFFFFFF
    v = philox4x32_10(np.full(nq, cursor & 0xFFFFFFFF), np.full(nq, cursor >> 32), q, np.full(nq, int(rank) & 0xFFFFFFFF), seed & 0xFFFFFFFF, seed >> 32)
    u = [((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0) for w in v]
    r0, r1 = np.sqrt(np.float32(-2) * np.log(u[0])), np.sqrt(np.float32(-2) * np.log(u[2]))
    t0, t1 = np.float32(6.28318530717958647692) * u[1], np.float32(6.28318530717958647692) * u[3]
    out = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], axis=1).astype(np.float32)
    return out.reshape(-1)[:n]


def _bf16(x):
    """round float32 to the nearest bfloat16 (ties to even), returned as float32 — what v_cvt_pk_bf16_f32 does"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def wfrag_image(Bm):
    """Bm[t][k][n] = B(k, n) of slab t (float32) -> the uint16 image [T][KK/16][ceil(NN/32)][3 terms][64 lanes][8] of its exact three-term
    bfloat16 split (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)); lane = 32 * (k // 8 % 2) + n % 32, element = k % 8"""
    T, KK, NN = Bm.shape
    JN = -(-NN // 32)
    x = np.zeros((T, KK, JN * 32), np.float32)
    x[:, :, :NN] = Bm
    h = _bf16(x)
    r = (x - h).astype(np.float32)
    m = _bf16(r)
    l = _bf16((r - m).astype(np.float32))
    planes = np.stack([(v.view(np.uint32) >> np.uint32(16)).astype(np.uint16) for v in (h, m, l)])      # [3][T][KK][JN*32]
    planes = planes.reshape(3, T, KK // 16, 2, 8, JN, 32)                                               # [plane][t][ks][hh][e][jn][j]
    return planes.transpose(1, 2, 5, 0, 3, 6, 4).reshape(-1)                                            # [t][ks][jn][plane][hh][j][e]


def stat_repl(C):
    """hp_stat_repl (include/hippie_hip.h): per-channel fp64 slots are double[R][2][C]; this interpreter adds into replica 0."""
    r = 2
    while r < STAT_REPL_MAX and r * 2 * C <= 1024:
        r *= 2
    return r


class Arenas:
    def __init__(self, sizes):
        self.mem = [np.zeros(int(s), dtype=np.uint8) for s in sizes]

    def view(self, ref, dtype, count):
        ref = int(ref)
        if ref == NULL:
            return None
        sp, off = ref >> 56, ref & MASK
        nb = np.dtype(dtype).itemsize * int(count)
        assert off + nb <= self.mem[sp].size, (sp, off, nb, self.mem[sp].size)
        return self.mem[sp][off: off + nb].view(dtype)

    def f32(self, ref, n):
        return self.view(ref, np.float32, n)

    def f64(self, ref, n):
        return self.view(ref, np.float64, n)

    def i64(self, ref, n):
        return self.view(ref, np.int64, n)


def _tapmap(i, conv=False):
    M, N, K, Lout, Lin, Pb, a, sh, _, nt = [int(v) for v in i[:10]]
    taps = [(int(i[10 + j]), int(i[16 + j]), int(i[22 + j]) if conv else 0) for j in range(nt)]
    return M, N, K, Lout, Lin, Pb, a, sh, taps


def _src_rows(M, Lout, Lin, Pb, a, sh, off):
    m = np.arange(M)
    b, l = m // Lout, m % Lout
    pos = a * l + off
    ok = (pos >= 0) & (pos < Pb)
    src = b * Lin + (np.where(ok, pos, 0) >> sh)
    return src, ok


def _out_rows(i, M, Lout):
    """output row of every GEMM row of a CONV_TAPS record, and the row count of the output tensor"""
    Lfull, oa, oo = int(i[28]), int(i[29]), int(i[30])
    m = np.arange(M)
    if Lfull == 0:
        return m, M
    return (m // Lout) * Lfull + oa * (m % Lout) + oo, (M // Lout) * Lfull


def _fma32(x, sc, sh):
    """fmaf(x, sc, sh) on float32 arrays (product exact in float64; the sum is rounded once to float64, then to float32)"""
    return (x.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float32)


def _torch_lerp(a, b, w):
    """torch.lerp: a + w*(b-a) for |w| < 0.5, else b - (b-a)*(1-w)."""
    d = b - a
    return (a + w * d if abs(float(w)) < 0.5 else b - d * (np.float32(1) - w)).astype(np.float32)


def _lrelu(x, s):
    return np.where(x > 0, x, x * np.float32(s)).astype(np.float32)


def _lrelu_grad(out, s):
    return np.where(out > 0, np.float32(1), np.float32(s)).astype(np.float32)


def _bn_coef(A, training, M, C, stats, gamma, beta, rmean, rvar, eps):
    if training:
        st = A.f64(stats, stat_repl(C) * 2 * C).reshape(stat_repl(C), 2 * C).sum(0)
        mean = st[:C] / M
        var = np.maximum(st[C:] / M - mean * mean, 0.0)
    else:
        mean = A.f32(rmean, C).astype(np.float64)
        var = A.f32(rvar, C).astype(np.float64)
    invstd = 1.0 / np.sqrt(var + np.float64(np.float32(eps)))
    g = A.f32(gamma, C).astype(np.float64)
    sc = g * invstd
    sh = A.f32(beta, C).astype(np.float64) - mean * sc
    return mean, var, invstd, sc.astype(np.float32), sh.astype(np.float32)


def _bn_dr(A, g, raw, save, bs_ref, gamma, dgamma, dbeta, C, Mstat, world):
    """HP_OP_BN_BWD_APPLY: dr = fma(g, A, fma(raw, B, C)) with the per-channel (A, B, C) of hp_common.h::bn_dr_coef; writes
    dgamma / dbeta.  Returns dr (float32)."""
    sv = A.f32(save, 2 * C).astype(np.float64)
    mean, invstd = sv[:C], sv[C:]
    bs = A.f64(bs_ref, stat_repl(C) * 2 * C).reshape(stat_repl(C), 2 * C).sum(0)
    sc = A.f32(gamma, C).astype(np.float64) * invstd
    c1, c2 = bs[:C] / Mstat, bs[C:] / Mstat
    cA, cB, cC = sc.astype(np.float32), (-sc * c2 * invstd).astype(np.float32), (sc * (c2 * invstd * mean - c1)).astype(np.float32)
    A.f32(dgamma, C)[:] = (bs[C:] / world).astype(np.float32)
    A.f32(dbeta, C)[:] = (bs[:C] / world).astype(np.float32)
    return _fma32(g, cA[None, :], _fma32(raw, cB[None, :], cC[None, :]))


def _bn_side(A, M, C, mean, var, invstd, save, rmean, rvar, mom, coef=NULL, sc=None, sh=None):
    sv = A.f32(save, 2 * C)
    sv[:C] = mean.astype(np.float32)
    sv[C:] = invstd.astype(np.float32)
    if int(coef) != NULL:
        cf = A.f32(coef, 2 * C)
        cf[:C], cf[C:] = sc, sh
    unb = var * M / (M - 1) if M > 1 else var
    rm, rv = A.f32(rmean, C), A.f32(rvar, C)
    mom = np.float64(np.float32(mom))
    rm[:] = ((1 - mom) * rm.astype(np.float64) + mom * mean.astype(np.float32).astype(np.float64)).astype(np.float32)
    rv[:] = ((1 - mom) * rv.astype(np.float64) + mom * unb).astype(np.float32)


def run(ops, A: Arenas, first=0, count=None):
    count = len(ops) - first if count is None else count
    for r in ops[first: first + count]:
        op, flags, i, f, b = int(r["op"]), int(r["flags"]), r["i"], r["f"], r["buf"]
        if op == 1:      # CONV_TAPS
            M, N, K, Lout, Lin, Pb, a, sh, taps = _tapmap(i, conv=True)
            nin = (M // Lout) * Lin
            orow, nout = _out_rows(i, M, Lout)
            acc = np.zeros((M, N), dtype=np.float64)
            coefs = None
            if flags & CONV_IN_BN:      # training-mode BatchNorm + leaky_relu of the input, evaluated in the loader
                Ms = int(i[31]) * max(1, int(i[32]))
                mean, var, invstd, sc_in, sh_in = _bn_coef(A, True, Ms, K, b[12], b[5], b[6], b[7], b[8], f[3])
                coefs = (sc_in, sh_in)
            for off, w, srcsel in taps:
                X = A.f32(b[10] if srcsel else b[0], nin * K).reshape(nin, K)
                W = A.f32(b[11] if srcsel else b[1], (w + 1) * N * K)
                src, ok = _src_rows(M, Lout, Lin, Pb, a, sh, off)
                Xs = X[src]
                if coefs is not None:
                    Xs = _lrelu(_fma32(Xs, coefs[0][None, :], coefs[1][None, :]), f[2])
                Ag = np.where(ok[:, None], Xs, 0)
                Ws = W[w * N * K: (w + 1) * N * K]
                if flags & CONV_BF16:
                    Ag, Ws = _bf16(Ag), _bf16(Ws)
                Ag = Ag.astype(np.float64)
                Wm = Ws.reshape(K, N) if flags & 1 else Ws.reshape(N, K).T
                acc += Ag @ Wm.astype(np.float64)
            if coefs is not None:       # the input BatchNorm's side effects (workgroup 0 of the kernel)
                _bn_side(A, Ms, K, mean, var, invstd, b[13], b[7], b[8], f[4], b[14], coefs[0], coefs[1])
            out = acc.astype(np.float32)
            dst = A.f32(b[2], nout * N).reshape(nout, N)
            if flags & CONV_EPI_BNRED:      # HP_OP_BN_BWD_REDUCE on the output
                raw = A.f32(b[17], nout * N).reshape(nout, N)[orow]
                if int(b[15]) != NULL:
                    out = out + A.f32(b[15], nout * N).reshape(nout, N)[orow]
                if int(b[16]) != NULL:
                    pre = A.f32(b[16], nout * N).reshape(nout, N)[orow]
                else:
                    cf = A.f32(b[19], 2 * N)
                    pre = _fma32(raw, cf[None, :N], cf[None, N:])
                g = (out * _lrelu_grad(pre, f[5])).astype(np.float32)
                dst[orow] = g
                for (raw_r, save_r, bs_r) in ([(b[17], b[18], b[20])] + ([(b[21], b[22], b[23])] if int(b[21]) != NULL else [])):
                    rw = A.f32(raw_r, nout * N).reshape(nout, N)[orow]
                    sv = A.f32(save_r, 2 * N)
                    xh = ((rw - sv[None, :N]) * sv[None, N:]).astype(np.float32)
                    bs = A.f64(bs_r, 2 * N)
                    bs[:N] += g.astype(np.float64).sum(0)
                    bs[N:] += (g.astype(np.float64) * xh.astype(np.float64)).sum(0)
                continue
            if flags & 2:
                out = out + A.f32(b[3], N)[None, :]
            if flags & 8:     # eval-mode BatchNorm (+ residual tensor, + leaky_relu with flag 16) folded into the epilogue
                _, _, _, sc, shf = _bn_coef(A, False, M, N, NULL, b[5], b[6], b[7], b[8], f[0])
                out = out * sc[None, :] + shf[None, :]
                if int(b[9]) != NULL:
                    out = out + A.f32(b[9], nout * N).reshape(nout, N)[orow]
                if flags & 16:
                    out = _lrelu(out, f[1])
                out = out.astype(np.float32)
            dst[orow] = out
            if flags & 4:
                st = A.f64(b[4], 2 * N)
                st[:N] += out.astype(np.float64).sum(0)
                st[N:] += (out.astype(np.float64) ** 2).sum(0)
        elif op == 2:    # WGRAD_TAPS
            M, N, K, Lout, Lin, Pb, a, sh, taps = _tapmap(i)
            nsplit, rps, stride = int(i[22]), int(i[23]), int(i[24])
            nin = (M // Lout) * Lin
            DY = A.f32(b[0], M * N).reshape(M, N).astype(np.float64)
            X = A.f32(b[1], nin * K).reshape(nin, K)
            if flags & CONV_IN_BN:
                cf = A.f32(b[3], 2 * K)
                X = _lrelu(_fma32(X, cf[None, :K], cf[None, K:]), f[0])
            if flags & CONV_BF16:
                DY, X = _bf16(DY.astype(np.float32)).astype(np.float64), _bf16(X)
            atomic = flags & 1
            slab = A.f32(b[2], stride if atomic else nsplit * stride)
            if not atomic:
                slab[:] = 0  # the kernel writes every tile of every split it owns
            for s in range(nsplit):
                lo, hi = s * rps, min(M, (s + 1) * rps)
                for off, w, _ in taps:
                    src, ok = _src_rows(M, Lout, Lin, Pb, a, sh, off)
                    Xg = np.where(ok[:, None], X[src], 0).astype(np.float64)
                    part = DY[lo:hi].T @ Xg[lo:hi]
                    if atomic:
                        slab[w * N * K: (w + 1) * N * K] += part.astype(np.float32).reshape(-1)
                    else:
                        slab[s * stride + w * N * K: s * stride + (w + 1) * N * K] = part.astype(np.float32).reshape(-1)
        elif op == 3:    # SLAB_REDUCE
            n, nsplit, stride = int(i[0]), int(i[1]), int(i[2])
            slab = A.f32(b[0], nsplit * stride)
            out = A.f32(b[1], n)
            acc = slab[:n].copy()
            for s in range(1, nsplit):
                acc = acc + slab[s * stride: s * stride + n]
            out[:] = acc
        elif op == 4:    # BN_APPLY
            M, C, res_mode, training, act = [int(v) for v in i[:5]]
            slope, eps, mom = f[0], f[1], f[2]
            raw = A.f32(b[0], M * C).reshape(M, C)
            Ms = M * max(1, int(i[5]))          # sync-BatchNorm: statistics summed over `world` ranks
            mean, var, invstd, sc, sh = _bn_coef(A, training, Ms, C, b[2], b[3], b[4], b[5], b[6], eps)
            v = raw * sc[None, :] + sh[None, :]
            if res_mode == 1:
                v = v + A.f32(b[8], M * C).reshape(M, C)
            elif res_mode == 2:
                mean2, var2, invstd2, sc2, sh2 = _bn_coef(A, training, Ms, C, b[9], b[10], b[11], b[12], b[13], eps)
                v = v + (A.f32(b[8], M * C).reshape(M, C) * sc2[None, :] + sh2[None, :])
            if act:
                v = _lrelu(v, slope)
            A.f32(b[1], M * C)[:] = v.astype(np.float32).reshape(-1)
            if training:
                _bn_side(A, Ms, C, mean, var, invstd, b[7], b[5], b[6], mom)
                if res_mode == 2:
                    _bn_side(A, Ms, C, mean2, var2, invstd2, b[14], b[12], b[13], mom)
        elif op == 5:    # BN_BWD_REDUCE
            M, C, has_g2, has_second = [int(v) for v in i[:4]]
            g = A.f32(b[0], M * C).reshape(M, C).copy()
            if has_g2:
                g = g + A.f32(b[1], M * C).reshape(M, C)
            if int(b[2]) != NULL:
                pre = A.f32(b[2], M * C).reshape(M, C)
            else:       # the activation was never stored: sign of fma(raw, scale, shift)
                cf = A.f32(b[10], 2 * C)
                pre = _fma32(A.f32(b[4], M * C).reshape(M, C), cf[None, :C], cf[None, C:])
            g = (g * _lrelu_grad(pre, f[0])).astype(np.float32)
            A.f32(b[3], M * C)[:] = g.reshape(-1)
            for (raw_r, save_r, bs_r) in ([(b[4], b[5], b[6])] + ([(b[7], b[8], b[9])] if has_second else [])):
                raw = A.f32(raw_r, M * C).reshape(M, C)
                sv = A.f32(save_r, 2 * C)
                xh = ((raw - sv[None, :C]) * sv[None, C:]).astype(np.float32)
                bs = A.f64(bs_r, 2 * C)
                bs[:C] += g.astype(np.float64).sum(0)
                bs[C:] += (g.astype(np.float64) * xh.astype(np.float64)).sum(0)
        elif op == 6:    # BN_BWD_APPLY
            M, C = int(i[0]), int(i[1])
            W = max(1, int(i[2]))
            g = A.f32(b[0], M * C).reshape(M, C)
            raw = A.f32(b[1], M * C).reshape(M, C)
            A.f32(b[5], M * C)[:] = _bn_dr(A, g, raw, b[2], b[3], b[4], b[6], b[7], C, M * W, W).reshape(-1)
        elif op in (7, 8):    # STEM_FWD / STEM_WGRAD
            B, Lin, Lout, C = [int(v) for v in i[:4]]
            xr = b[0] if op == 7 else b[1]
            x = A.f32(xr, B * Lin).reshape(B, Lin)
            xp = np.pad(x, ((0, 0), (1, 2)))
            cols = np.stack([xp[:, 2 * np.arange(Lout) + t] for t in range(3)], axis=-1)   # [B, Lout, 3]
            if op == 7:
                W = A.f32(b[1], C * 3).reshape(C, 3)
                out = (cols.reshape(B * Lout, 3).astype(np.float64) @ W.T.astype(np.float64)).astype(np.float32)
                A.f32(b[2], B * Lout * C)[:] = out.reshape(-1)
                if int(b[3]) != NULL:
                    st = A.f64(b[3], 2 * C)
                    st[:C] += out.astype(np.float64).sum(0)
                    st[C:] += (out.astype(np.float64) ** 2).sum(0)
            else:
                dr = A.f32(b[0], B * Lout * C).reshape(B * Lout, C).astype(np.float64)
                A.f32(b[2], C * 3)[:] += (dr.T @ cols.reshape(B * Lout, 3).astype(np.float64)).astype(np.float32).reshape(-1)
        elif op == 9:    # POOL_FWD
            B, L, C = [int(v) for v in i[:3]]
            A.f32(b[1], B * C)[:] = (A.f32(b[0], B * L * C).reshape(B, L, C).sum(1) / np.float32(L)).reshape(-1)
        elif op == 10:   # POOL_BWD
            B, L, C = [int(v) for v in i[:3]]
            d = A.f32(b[0], B * C).reshape(B, 1, C) / np.float32(L)
            A.f32(b[1], B * L * C)[:] = np.broadcast_to(d, (B, L, C)).reshape(-1)
        elif op == 11:   # REPEAT_FWD
            B, R, C = [int(v) for v in i[:3]]
            A.f32(b[1], B * R * C)[:] = np.broadcast_to(A.f32(b[0], B * C).reshape(B, 1, C), (B, R, C)).reshape(-1)
        elif op == 12:   # REPEAT_BWD
            B, R, C, has_g2 = [int(v) for v in i[:4]]
            g = A.f32(b[0], B * R * C).reshape(B, R, C).copy()
            if has_g2:
                g = g + A.f32(b[1], B * R * C).reshape(B, R, C)
            A.f32(b[2], B * C)[:] = g.sum(1).reshape(-1)
        elif op == 13:   # CONCAT
            B, nseg, ldo = int(i[0]), int(i[1]), int(i[2])
            out = A.f32(b[0], B * ldo).reshape(B, ldo)
            col = 0
            for j in range(nseg):
                kind, w, ld = int(i[4 + 3 * j]), int(i[5 + 3 * j]), int(i[6 + 3 * j])
                if kind == 0:
                    out[:, col: col + w] = A.f32(b[1 + 2 * j], B * ld).reshape(B, ld)[:, :w]
                elif kind == 1:
                    idx = A.i64(b[2 + 2 * j], B)
                    rows = int(i[16 + j])
                    ok = (idx >= 0) & (idx < rows)           # out-of-range label: zero row (device-side guard)
                    tab = A.f32(b[1 + 2 * j], rows * ld).reshape(rows, ld)
                    out[:, col: col + w] = np.where(ok[:, None], tab[np.where(ok, idx, 0)][:, :w], np.float32(0))
                else:
                    out[:, col: col + w] = 0
                col += w
            assert col == ldo
        elif op == 14:   # EMB_BWD
            B, w, ld, col0, rows = [int(v) for v in i[:5]]
            d = A.f32(b[0], B * ld).reshape(B, ld)[:, col0: col0 + w]
            idx = A.i64(b[1], B)
            ok = (idx >= 0) & (idx < rows)                   # out-of-range label: skipped
            dt = A.f32(b[2], rows * w).reshape(rows, w)
            np.add.at(dt, idx[ok], d[ok])
        elif op == 15:   # LINEAR_FWD
            M, N, K, ldx, ldy, act, stats = [int(v) for v in i[:7]]
            X = A.f32(b[0], (M - 1) * ldx + K)
            Xm = np.lib.stride_tricks.as_strided(X, (M, K), (ldx * 4, 4))
            W = A.f32(b[1], N * K).reshape(N, K)
            y = (Xm.astype(np.float64) @ W.T.astype(np.float64)).astype(np.float32)
            if int(b[2]) != NULL:
                y = y + A.f32(b[2], N)[None, :]
            if stats:
                st = A.f64(b[4], 2 * N)
                st[:N] += y.astype(np.float64).sum(0)
                st[N:] += (y.astype(np.float64) ** 2).sum(0)
            if act:
                y = _lrelu(y, f[0])
            Y = A.f32(b[3], (M - 1) * ldy + N)
            np.lib.stride_tricks.as_strided(Y, (M, N), (ldy * 4, 4))[:] = y
        elif op == 16:   # LINEAR_BWD_X
            M, N, K, ldy, ldx, has_mask, lda, accum = [int(v) for v in i[:8]]
            DY = np.lib.stride_tricks.as_strided(A.f32(b[0], (M - 1) * ldy + N), (M, N), (ldy * 4, 4))
            W = A.f32(b[1], N * K).reshape(N, K)
            dx = (DY.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)
            if has_mask:
                act = np.lib.stride_tricks.as_strided(A.f32(b[3], (M - 1) * lda + K), (M, K), (lda * 4, 4))
                dx = dx * _lrelu_grad(act, f[0])
            DX = np.lib.stride_tricks.as_strided(A.f32(b[2], (M - 1) * ldx + K), (M, K), (ldx * 4, 4))
            if accum:
                DX[:] = DX + dx
            else:
                DX[:] = dx
        elif op == 17:   # LINEAR_BWD_W
            M, N, K, ldy, ldx = [int(v) for v in i[:5]]
            DY = np.lib.stride_tricks.as_strided(A.f32(b[0], (M - 1) * ldy + N), (M, N), (ldy * 4, 4)).astype(np.float64)
            X = np.lib.stride_tricks.as_strided(A.f32(b[1], (M - 1) * ldx + K), (M, K), (ldx * 4, 4)).astype(np.float64)
            A.f32(b[2], N * K)[:] += (DY.T @ X).astype(np.float32).reshape(-1)
            if int(b[3]) != NULL:
                A.f32(b[3], N)[:] += DY.sum(0).astype(np.float32)
        elif op == 18:   # REPARAM_KL_FWD
            B, z = int(i[0]), int(i[1])
            mulv = A.f32(b[0], B * 2 * z).reshape(B, 2 * z)
            mu, lv = mulv[:, :z], mulv[:, z:]
            eps = A.f32(b[1], B * z).reshape(B, z)
            A.f32(b[2], B * z)[:] = (mu + eps * np.exp(np.float32(0.5) * lv)).astype(np.float32).reshape(-1)
            A.f64(b[3], 4)[0] += float((-0.5 * (1 + lv - mu * mu - np.exp(lv)).astype(np.float64)).sum())
        elif op == 19:   # REPARAM_KL_BWD
            B, z, ld = int(i[0]), int(i[1]), int(i[2])
            beta = np.float32(f[0])
            mulv = A.f32(b[0], B * 2 * z).reshape(B, 2 * z)
            mu, lv = mulv[:, :z], mulv[:, z:]
            eps = A.f32(b[1], B * z).reshape(B, z)
            dz = np.lib.stride_tricks.as_strided(A.f32(b[2], (B - 1) * ld + z), (B, z), (ld * 4, 4))
            out = A.f32(b[3], B * 2 * z).reshape(B, 2 * z)
            out[:, :z] = dz + beta * mu / np.float32(B)
            out[:, z:] = dz * eps * np.float32(0.5) * np.exp(np.float32(0.5) * lv) + beta * np.float32(0.5) * (np.exp(lv) - 1) / np.float32(B)
        elif op == 20:   # MSE_FWD_BWD
            n, slot = int(i[0]), int(i[1])
            x, rec = A.f32(b[0], n), A.f32(b[1], n)
            d = rec - x
            A.f64(b[3], 4)[slot] += float((d.astype(np.float64) ** 2).sum())
            A.f32(b[2], n)[:] = np.float32(f[0]) * np.float32(2) * d / np.float32(n)
        elif op in (21, 22, 23):   # TAIL ops
            B, Lh, C = [int(v) for v in i[:3]]
            Lo = 2 * Lh
            if op == 21:
                act = A.f32(b[0], B * Lh * C).reshape(B, Lh, C)
                W = A.f32(b[1], C * 3).reshape(C, 3).astype(np.float64)
                up = np.pad(np.repeat(act, 2, axis=1), ((0, 0), (1, 1), (0, 0))).astype(np.float64)   # [B, Lo+2, C]
                out = sum(up[:, t: t + Lo, :] @ W[:, t] for t in range(3)) + float(A.f32(b[2], 1)[0])
                A.f32(b[3], B * Lo)[:] = out.astype(np.float32).reshape(-1)
            elif op == 22:
                dt = np.pad(A.f32(b[0], B * Lo).reshape(B, Lo), ((0, 0), (2, 2))).astype(np.float64)
                W = A.f32(b[1], C * 3).reshape(C, 3).astype(np.float64)
                dup = sum(dt[:, 2 + 1 - t: 2 + 1 - t + Lo, None] * W[None, None, :, t] for t in range(3))   # [B, Lo, C]
                A.f32(b[2], B * Lh * C)[:] = (dup[:, 0::2] + dup[:, 1::2]).astype(np.float32).reshape(-1)
            else:
                dt = A.f32(b[0], B * Lo).reshape(B, Lo).astype(np.float64)
                act = A.f32(b[1], B * Lh * C).reshape(B, Lh, C)
                up = np.pad(np.repeat(act, 2, axis=1), ((0, 0), (1, 1), (0, 0))).astype(np.float64)
                dw = np.stack([np.einsum("bp,bpc->c", dt, up[:, t: t + Lo, :]) for t in range(3)], axis=1)   # [C,3]
                A.f32(b[2], C * 3)[:] += dw.astype(np.float32).reshape(-1)
                A.f32(b[3], 1)[0] += np.float32(dt.sum())
        elif op == 24:   # LOSS_FINALIZE
            B, n1, n2 = int(i[0]), int(i[1]), int(i[2])
            L = A.f64(b[0], 4)
            kl, m1 = L[0] / B, L[1] / n1
            m2 = L[2] / n2 if n2 > 0 else 0.0
            out = A.f32(b[1], 4)
            out[0] = np.float32(float(f[1]) * m1 + float(f[2]) * m2 + float(f[0]) * kl)
            out[1], out[2], out[3] = np.float32(m1), np.float32(m2), np.float32(kl)
        elif op == 25:   # GRADNORM
            n = int(i[0])
            A.f64(b[1], 1)[0] += float((A.f32(b[0], n).astype(np.float64) ** 2).sum())
        elif op == 26:   # ADAMW
            n = int(i[0])
            lr, b1, b2, eps, wd, clip, omb1, omb2 = [np.float32(v) for v in f[:8]]
            p, g, m, v = A.f32(b[0], n), A.f32(b[1], n), A.f32(b[2], n), A.f32(b[3], n)
            t = float(A.i64(b[4], 1)[0])
            bc1 = 1.0 - (1.0 - float(omb1)) ** t
            bc2 = 1.0 - (1.0 - float(omb2)) ** t
            coef = np.float32(1)
            if clip > 0:
                c = clip / (np.float32(np.sqrt(A.f64(b[5], 1)[0])) + np.float32(1e-6))
                coef = min(c, np.float32(1))
            gg = (g * coef).astype(np.float32)
            p[:] = p * (np.float32(1) - lr * wd)
            m[:] = m + omb1 * (gg - m)
            v[:] = v * b2 + omb2 * gg * gg
            denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + eps
            p[:] = p - np.float32(float(lr) / bc1) * (m / denom)
        elif op == 27:   # STEP_INC
            A.i64(b[0], 1)[0] += 1
        elif op == 28:   # ZERO
            nb = int(np.uint32(i[0])) + (int(np.uint32(i[1])) << 32)
            A.view(b[0], np.uint8, nb)[:] = 0
        elif op == 32:   # SF_SCHEDULE (hippie/optimizers.py:118-138)
            k = float(A.i64(b[0], 1)[0])
            st = A.f64(b[1], 4)
            warm = int(i[0])
            lr, omb2, r, power = [float(np.float32(v)) for v in f[:4]]
            sched = (k + 1) / warm if k < warm else 1.0
            lr_t = lr * sched * np.sqrt(1.0 - (1.0 - omb2) ** (k + 1))
            st[0] = max(lr_t, st[0])
            weight = (k + 1) ** r * st[0] ** power
            st[1] += weight
            st[2] = lr_t
            st[3] = weight / st[1] if st[1] != 0 else 0.0
        elif op == 33:   # ADAMW_SF (hippie/optimizers.py:145-207)
            n = int(i[0])
            b1, b2, eps, wd, clip, omb2 = [np.float32(v) for v in f[:6]]
            y, g, z, v = A.f32(b[0], n), A.f32(b[1], n), A.f32(b[2], n), A.f32(b[3], n)
            st = A.f64(b[5], 4)
            if int(A.i64(b[4], 1)[0]) == 0:
                z[:] = y
            coef = np.float32(1)
            if clip > 0:
                c = clip / (np.float32(np.sqrt(A.f64(b[6], 1)[0])) + np.float32(1e-6))
                coef = min(c, np.float32(1))
            gg = (g * coef).astype(np.float32)
            v[:] = v * b2 + omb2 * gg * gg
            gn = gg / (np.sqrt(v) + eps)
            if wd != 0:
                gn = gn + wd * y
            y[:] = _torch_lerp(y, z, np.float32(st[3]))
            y[:] = y + np.float32(st[2] * (float(b1) * (1.0 - st[3]) - 1.0)) * gn
            z[:] = z - np.float32(st[2]) * gn
        elif op == 34:   # LERP
            n = int(i[0])
            y = A.f32(b[0], n)
            y[:] = _torch_lerp(y, A.f32(b[1], n), np.float32(f[0]))
        elif op == 31:   # RESAMPLE_LINEAR
            from oracle.preproc import resample_linear
            N, W, L = int(i[0]), int(i[1]), int(i[2])
            x = A.f32(b[0], N * W).reshape(N, W)
            if flags & 1:
                x = np.log(x + np.float32(1)).astype(np.float32)
            A.f32(b[1], N * L)[:] = resample_linear(x, L).reshape(-1)
        elif op == 35:         # STATS_SYNC: single-process no-op (the host sums the slot over ranks here)
            pass
        elif op == 36:         # STAGE_BATCH: index gather from resident tables + Philox noise
            B, L, L2, z, spe, world, rank, N = [int(v) for v in i[:8]]
            cur = int(A.i64(b[4], 1)[0])
            j = (cur % spe) * world + rank
            rows = A.i64(b[3], spe * world * B)[j * B: (j + 1) * B].copy()
            rows[(rows < 0) | (rows >= N)] = 0
            A.f32(b[5], B * L)[:] = A.f32(b[0], N * L).reshape(N, L)[rows].reshape(-1)
            if L2 > 0 and int(b[1]) != NULL:
                A.f32(b[6], B * L2)[:] = A.f32(b[1], N * L2).reshape(N, L2)[rows].reshape(-1)
            A.i64(b[7], B)[:] = A.i64(b[2], N)[rows]
            A.f32(b[8], B * z)[:] = philox_normal(int(A.i64(b[9], 1)[0]), cur, B * z, rank)
        elif op in (29, 30, 37):   # WGRAD_GROUP / PAIR / HEADS: their member records (before them) were executed in place
            pass
        elif op == 38:         # WFRAG: three-term MFMA fragments of a conv weight tensor (include/hippie_hip.h, HP_OP_WFRAG)
            T, N, K, which = [int(v) for v in i[:4]]
            W = A.f32(b[0], T * N * K).reshape(T, N, K)
            if which & 1:
                A.view(b[1], np.uint16, T * (K // 16) * (-(-N // 32)) * 1536)[:] = wfrag_image(W.transpose(0, 2, 1))
            if which & 2:
                A.view(b[2], np.uint16, T * (N // 16) * (K // 32) * 1536)[:] = wfrag_image(W)
        else:
            raise ValueError(f"unknown opcode {op}")
