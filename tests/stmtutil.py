"""Inputs of a mixed-type Statement (statement.rs:24-28,109) for the GPU parity tests: ONE list of variables -- G1
variables xg, G2 variables yg, scalar variables xs, ys, all with known discrete logarithms -- and equations of the four
types over them with SATISFIED targets (the generator knows the logarithms, as workload.py does for batches).
Test plumbing only: group elements are made with the engine's own scalar-multiplication / pairing helpers."""
import numpy as np

from groth_sahai_rs_amd.workload import CURVES, Workload, _limbs


class StatementInputs:
    def __init__(self, eng, mg=3, ng=2, ms=2, ns=3, seed=5150):
        import torch

        self.eng = eng
        # a CRS of the generator.rs shape (and the generators p1, p2, gt = e(p1, p2)) through the batch workload's code
        wl = Workload(eng, ty=0, N=1, m=1, n=1, seed=seed, corrupt_every=0)
        self.crs, self.p1, self.p2, self.gt = wl.crs, wl.g1_gen, wl.g2_gen, wl.gt_gen
        self.r = CURVES[eng.curve]["r"]
        self.rng = np.random.default_rng(seed)
        self.torch = torch
        self.dlog = dict(xg=self.scalars(mg), yg=self.scalars(ng), xs=self.scalars(ms), ys=self.scalars(ns))
        self.vars = dict(xg=self.g(1, self.dlog["xg"]), yg=self.g(2, self.dlog["yg"]), xs=self.fr(self.dlog["xs"]),
                         ys=self.fr(self.dlog["ys"]))
        self.rand = dict(xg=self.fr(self.scalars(2 * mg)), yg=self.fr(self.scalars(2 * ng)), xs=self.fr(self.scalars(ms)),
                         ys=self.fr(self.scalars(ns)))

    def scalars(self, count):
        raw = self.rng.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
        return [(int(a) | int(b) << 64 | int(c) << 128 | int(d) << 192) % self.r for a, b, c, d in raw]

    def fr(self, vals):
        """Montgomery-form scalars as bytes"""
        a = np.array([_limbs(v * (1 << 256) % self.r, 4) for v in vals], dtype=np.uint64).reshape(-1)
        return a.view(np.uint8)

    def g(self, group, vals):
        if not vals:
            return np.zeros(0, dtype=np.uint8)
        base = self.p1 if group == 1 else self.p2
        return self.eng.g_mul_batch(group, base, self.fr(vals).view(np.uint64), broadcast=True).reshape(-1)

    @staticmethod
    def groups(ty):
        return ("xg" if ty in (0, 1) else "xs"), ("yg" if ty in (0, 2) else "ys")

    def part(self, ty, E, satisfied=True):
        """E equations of type `ty` over the statement's variables: dict of host arrays A, B, Gamma, target, T plus the
        shared X, Y, R, S of the groups the type uses."""
        gx, gy = self.groups(ty)
        xd, yd = self.dlog[gx], self.dlog[gy]
        m, n, r = len(xd), len(yd), self.r
        a, b, gam, tg = self.scalars(E * n), self.scalars(E * m), self.scalars(E * m * n), []
        for e in range(E):
            s = sum(a[e * n + j] * yd[j] for j in range(n)) + sum(xd[i] * b[e * m + i] for i in range(m))
            s += sum(xd[i] * gam[(e * m + i) * n + j] % r * yd[j] for i in range(m) for j in range(n))
            tg.append((s + (0 if satisfied else 1 + e)) % r)
        xg, yg = ty in (0, 1), ty in (0, 2)
        A = self.g(1, a) if xg else self.fr(a)
        B = self.g(2, b) if yg else self.fr(b)
        if ty == 0:
            T = self.torch
            kt = T.from_numpy(self.fr(tg).copy()).to("cuda:0")
            out = T.empty(E * self.eng.GT, dtype=T.uint8, device="cuda:0")
            self.eng.gt_pow_batch_dev(E, T.from_numpy(self.gt.copy()).to("cuda:0"), kt, out)
            self.eng.sync()
            target = out.cpu().numpy()
        elif ty == 1:
            target = self.g(1, tg)
        elif ty == 2:
            target = self.g(2, tg)
        else:
            target = self.fr(tg)
        kx, ky = (2 if xg else 1), (2 if yg else 1)
        return dict(ty=ty, N=E, m=m, n=n, shared=True, X=self.vars[gx], Y=self.vars[gy], R=self.rand[gx], S=self.rand[gy],
                    A=A, B=B, Gamma=self.fr(gam), T=self.fr(self.scalars(E * ky * kx)), target=target)
