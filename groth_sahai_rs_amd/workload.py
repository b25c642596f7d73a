"""Synthetic Groth-Sahai batches, generated ON the GPU through
the engine's own helper kernels so that 2^12..2^18-equation workloads take
seconds to build.  Shape of the reference's bench inputs (benches/bench.rs:420-442,
500-522): one CRS of the generator.rs:81-118 shape, witnesses and constants as
uniform multiples of the CRS generators, dense uniform Gamma, SATISFIED targets
(the generator knows the discrete logs), every `corrupt_every`-th proof gets one
flipped bit and must be rejected.

Everything here is input plumbing for bench.py / tests; it contains no CPU
implementation of the hot path.
"""
import numpy as np

from .capi import GS_MSMEG1, GS_MSMEG2, GS_PPE, GS_QUAD

# standard generators / moduli (canonical integers) -- public curve constants
CURVES = {
    0: dict(
        name="bls12_381",
        p=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
        r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
        g1=(0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
            0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
        g2=(0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
    ),
    1: dict(
        name="bn254",
        p=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        g1=(1, 2),
        g2=(10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531),
    ),
}


class SplitMix64:
    """splitmix64 (seed = 20241220 + config index)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)


def _limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


class Workload:
    """Device-resident batch of N equations of one type/shape plus its CRS."""

    def __init__(self, eng, ty=GS_PPE, N=4096, m=4, n=4, seed=20241220, corrupt_every=1024, device="cuda:0"):
        import torch

        self.eng, self.ty, self.N, self.m, self.n = eng, ty, N, m, n
        cv = CURVES[eng.curve]
        self.p, self.r = cv["p"], cv["r"]
        nq = eng.FQ // 8
        Rq, Rr = 1 << (64 * nq), 1 << 256
        self.nq = nq
        sh = eng.shape(ty)
        self.sh = sh
        kx, ky = sh["kx"], sh["ky"]
        rng = np.random.default_rng(seed)
        sm = SplitMix64(seed)
        r = self.r

        def rand_fr(count):
            """count uniform scalars as python ints < r"""
            raw = rng.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
            out = []
            for row in raw:
                v = (int(row[0]) | int(row[1]) << 64 | int(row[2]) << 128 | int(row[3]) << 192) % r
                out.append(v)
            return out

        def fr_tensor(vals):
            a = np.array([_limbs(v * Rr % r, 4) for v in vals], dtype=np.uint64)
            return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device)

        def fq_bytes(v):
            return np.array(_limbs(v * Rq % self.p, nq), dtype=np.uint64).view(np.uint8)

        def raw_fr_tensor(count):
            """uniform scalars given directly as Montgomery limbs (as arkworks' Fr::rand does)"""
            raw = rng.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
            top = np.uint64((1 << (r.bit_length() - 192)) - 1)
            raw[:, 3] &= top
            # rejection: compare with r as 4 limbs; resample the (rare) rows >= r
            rl = _limbs(r, 4)
            for i in range(count):
                while True:
                    v = int(raw[i, 0]) | int(raw[i, 1]) << 64 | int(raw[i, 2]) << 128 | int(raw[i, 3]) << 192
                    if v < r:
                        break
                    raw[i] = rng.integers(0, 1 << 64, size=4, dtype=np.uint64)
                    raw[i, 3] &= top
            del rl
            return torch.from_numpy(raw.view(np.uint8).reshape(-1)).to(device)

        # ---- CRS (generator.rs:81-118 shape) built with the engine's own kernels
        g1std = np.concatenate([fq_bytes(cv["g1"][0]), fq_bytes(cv["g1"][1])])
        g2std = np.concatenate([fq_bytes(v) for v in cv["g2"]])
        al, be, a1, a2, t1, t2 = [sm.next() | (sm.next() << 64) | (sm.next() << 128) for _ in range(6)]
        al, be, a1, a2, t1, t2 = [v % r for v in (al, be, a1, a2, t1, t2)]
        frh = lambda vals: np.array([_limbs(v * Rr % r, 4) for v in vals], dtype=np.uint64)
        p1 = eng.g_mul_batch(1, g1std, frh([al]), broadcast=True)[0]
        p2 = eng.g_mul_batch(2, g2std, frh([be]), broadcast=True)[0]
        g1s = eng.g_mul_batch(1, p1, frh([a1, t1, a1 * t1 % r]), broadcast=True)  # q1, u1, v1 = t1*q1
        g2s = eng.g_mul_batch(2, p2, frh([a2, t2, a2 * t2 % r]), broadcast=True)
        gt = eng.multi_pairing_batch(1, 1, p1, p2)[0]
        self.crs = np.concatenate([p1, g1s[0], g1s[1], g1s[2], p2, g2s[0], g2s[1], g2s[2], p1, p2, gt])
        eng.set_crs(self.crs)
        self.g1_gen, self.g2_gen, self.gt_gen = p1, p2, gt

        # ---- per-equation scalars (discrete logs) and satisfied targets
        xs, ys = rand_fr(N * m), rand_fr(N * n)
        as_, bs = rand_fr(N * n), rand_fr(N * m)
        gam = rand_fr(N * m * n)
        tg = []
        for e in range(N):
            s = 0
            for j in range(n):
                s += as_[e * n + j] * ys[e * n + j]
            for i in range(m):
                s += xs[e * m + i] * bs[e * m + i]
                xi = xs[e * m + i]
                for j in range(n):
                    s += xi * gam[(e * m + i) * n + j] % r * ys[e * n + j]
            tg.append(s % r)
        T = torch
        dev_g1 = T.from_numpy(p1.copy()).to(device)
        dev_g2 = T.from_numpy(p2.copy()).to(device)
        dev_gt = T.from_numpy(gt.copy()).to(device)

        def group_or_fr(vals, group, isg):
            k = fr_tensor(vals)
            if not isg:
                return k
            out = T.empty(len(vals) * (eng.G1 if group == 1 else eng.G2), dtype=T.uint8, device=device)
            eng.g_mul_batch_dev(group, len(vals), dev_g1 if group == 1 else dev_g2, True, k, out)
            return out

        self.X = group_or_fr(xs, 1, sh["xg"])
        self.A = group_or_fr(as_, 1, sh["xg"])
        self.Y = group_or_fr(ys, 2, sh["yg"])
        self.B = group_or_fr(bs, 2, sh["yg"])
        self.Gamma = fr_tensor(gam)
        kt = fr_tensor(tg)
        if ty == GS_PPE:
            self.target = T.empty(N * eng.GT, dtype=T.uint8, device=device)
            eng.gt_pow_batch_dev(N, dev_gt, kt, self.target)
        elif ty == GS_MSMEG1:
            self.target = T.empty(N * eng.G1, dtype=T.uint8, device=device)
            eng.g_mul_batch_dev(1, N, dev_g1, True, kt, self.target)
        elif ty == GS_MSMEG2:
            self.target = T.empty(N * eng.G2, dtype=T.uint8, device=device)
            eng.g_mul_batch_dev(2, N, dev_g2, True, kt, self.target)
        else:
            self.target = kt
        self.R = raw_fr_tensor(N * m * kx)
        self.S = raw_fr_tensor(N * n * ky)
        self.T = raw_fr_tensor(N * ky * kx)
        # outputs
        self.xcoms = T.empty(N * m * eng.COM1, dtype=T.uint8, device=device)
        self.ycoms = T.empty(N * n * eng.COM2, dtype=T.uint8, device=device)
        self.pi = T.empty(N * kx * eng.COM2, dtype=T.uint8, device=device)
        self.theta = T.empty(N * ky * eng.COM1, dtype=T.uint8, device=device)
        self.ok = T.empty(N, dtype=T.uint8, device=device)
        self.corrupt_every = corrupt_every
        eng.sync()
        T.cuda.synchronize()

    # algorithmic bytes per prove+verify unit (SURVEY.md 8d formulae, generalised to the type's element sizes)
    def bytes_per_unit(self):
        e, m, n, sh = self.eng, self.m, self.n, self.sh
        kx, ky = sh["kx"], sh["ky"]
        rd_p = m * sh["sx"] + n * sh["sy"] + n * sh["sx"] + m * sh["sy"] + 32 * (m * n + m * kx + n * ky + kx * ky)
        wr_p = m * e.COM1 + n * e.COM2 + kx * e.COM2 + ky * e.COM1
        rd_v = n * sh["sx"] + m * sh["sy"] + 32 * m * n + sh["st"] + wr_p
        return rd_p + wr_p + rd_v + 1

    def prove(self):
        self.eng.prove_batch_dev(self.ty, self.N, self.m, self.n, self.X, self.Y, self.A, self.B, self.Gamma, self.R,
                                 self.S, self.T, self.xcoms, self.ycoms, self.pi, self.theta)

    def corrupt(self):
        """flip one bit of pi for every corrupt_every-th proof (must then be rejected)"""
        if not self.corrupt_every:
            return []
        idx = list(range(self.corrupt_every - 1, self.N, self.corrupt_every))
        stride = self.sh["kx"] * self.eng.COM2
        for i in idx:
            self.pi[i * stride + 5] ^= 4
        return idx

    def verify(self):
        self.eng.verify_batch_dev(self.ty, self.N, self.m, self.n, self.A, self.B, self.Gamma, self.target,
                                  self.xcoms, self.ycoms, self.pi, self.theta, self.ok)

    def verify_rlc(self):
        """batched verifier: ONE final exponentiation for the batch; returns the accumulator pair (device)"""
        import torch

        import os

        import numpy as np

        # the rho contract of gs_amd.h: fresh per call, from the OS CSPRNG, non-zero, drawn after the proofs exist
        raw = np.frombuffer(os.urandom(self.N * 32), dtype=np.uint64).copy()
        raw[raw == 0] = 1
        self.rho = torch.from_numpy(raw.view(np.int64)).to(self.xcoms.device)
        if not hasattr(self, "acc"):
            self.acc = torch.empty(2 * self.eng.GT, dtype=torch.uint8, device=self.xcoms.device)
        self.eng.verify_batch_rlc_dev(self.ty, self.N, self.m, self.n, self.A, self.B, self.Gamma, self.target,
                                      self.xcoms, self.ycoms, self.pi, self.theta, self.rho, self.acc)
        return self.acc

    def step(self):
        """one pass of the hot path over the batch: commit_and_prove then verify"""
        self.prove()
        self.verify()
