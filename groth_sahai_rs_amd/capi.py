"""ctypes binding of include/gs_amd.h.  Buffers are numpy arrays (host entry
points) or torch CUDA tensors (``*_dev`` entry points, zero-copy via data_ptr)."""
import ctypes
import os

import numpy as np

GS_PPE, GS_MSMEG1, GS_MSMEG2, GS_QUAD = 0, 1, 2, 3
CURVE_BLS12_381, CURVE_BN254 = 0, 1
_ERR = {1: "GS_ERR_SHAPE", 2: "GS_ERR_DEVICE", 3: "GS_ERR_ARG", 4: "GS_ERR_NOCRS", 5: "GS_ERR_ALLOC"}

_HERE = os.path.dirname(os.path.abspath(__file__))

# every symbol include/gs_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "gs_ctx_create", "gs_ctx_destroy", "gs_set_stream", "gs_set_option", "gs_sync", "gs_host_register",
    "gs_host_alloc", "gs_host_free",
    "gs_host_unregister", "gs_last_error", "gs_version", "gs_sizes",
    "gs_set_crs", "gs_crs_generate", "gs_crs_generate_hiding",
    "gs_commit_g1_dev", "gs_commit_g2_dev", "gs_commit_fr_b1_dev", "gs_commit_fr_b2_dev",
    "gs_commit_g1", "gs_commit_g2", "gs_commit_fr_b1", "gs_commit_fr_b2",
    "gs_prove_batch_dev", "gs_prove_batch", "gs_verify_batch_dev", "gs_verify_batch",
    "gs_verify_batch_rlc_dev", "gs_verify_batch_rlc", "gs_gt_finalize",
    "gs_prove_statement_dev", "gs_prove_statement", "gs_verify_statement_dev", "gs_verify_statement",
    "gs_mat_left_mul_com1", "gs_mat_left_mul_com2", "gs_pairing_sum", "gs_fr_matmul",
    "gs_g1_mul_batch", "gs_g2_mul_batch", "gs_g1_mul_batch_dev", "gs_g2_mul_batch_dev",
    "gs_multi_pairing_batch", "gs_multi_pairing_batch_dev", "gs_gt_pow_batch_dev",
    "gs_wire_sizes", "gs_wire_encode_g1", "gs_wire_encode_g2", "gs_wire_decode_g1", "gs_wire_decode_g2",
    "gs_wire_encode_fr", "gs_wire_decode_fr", "gs_wire_encode_gt", "gs_wire_decode_gt",
    "gs_validate_points", "gs_validate_points_dev",
    "gs_prof_enable", "gs_prof_reset", "gs_prof_get", "gs_prof_get_work", "gs_prof_get_clock",
    "gs_ctx_create_multi", "gs_multi_destroy", "gs_multi_ndev", "gs_multi_ctx", "gs_multi_last_error",
    "gs_multi_uses_rccl", "gs_multi_shard", "gs_multi_set_crs", "gs_multi_prove_batch", "gs_multi_verify_batch",
    "gs_multi_verify_batch_rlc",
    "gs_ctx_create_multi_ex", "gs_multi_exchange_note", "gs_multi_sync", "gs_multi_set_option", "gs_multi_prove_batch_dev",
    "gs_multi_verify_batch_dev", "gs_multi_verify_batch_rlc_dev", "gs_gt_finalize_dev",
    "gs_prove_mixed_dev", "gs_prove_mixed", "gs_verify_mixed_dev", "gs_verify_mixed",
]
GS_MIXED_MAX = 8
GS_MULTI_SHARED_DEVICES = 1


class ProvePart(ctypes.Structure):  # gs_prove_part
    _fields_ = [("equ_type", ctypes.c_int), ("N", ctypes.c_size_t), ("m", ctypes.c_int), ("n", ctypes.c_int)] + \
               [(k, ctypes.c_void_p) for k in ("X", "Y", "A", "B", "Gamma", "R", "S", "T", "xcoms", "ycoms", "pi", "theta")] + \
               [("shared_vars", ctypes.c_int)]


class VerifyPart(ctypes.Structure):  # gs_verify_part
    _fields_ = [("equ_type", ctypes.c_int), ("N", ctypes.c_size_t), ("m", ctypes.c_int), ("n", ctypes.c_int)] + \
               [(k, ctypes.c_void_p) for k in ("A", "B", "Gamma", "target", "xcoms", "ycoms", "pi", "theta", "ok")] + \
               [("shared_vars", ctypes.c_int)]


class GsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_ERR.get(code, "GS_ERR"), code, msg))
        self.code = code


def lib_path():
    # GS_AMD_LIB: load another build of the same library (A/B measurements); it is still a HIP build, never a fallback
    return os.environ.get("GS_AMD_LIB") or os.path.join(_HERE, "lib", "libgs_amd.so")


_LIB = None


def load_library():
    """Load the HIP extension; fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError(
                "groth_sahai_rs_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p
            )
        try:
            # PyTorch-ROCm bundles its own HIP runtime; when both end up in one process torch's copy must be
            # loaded first, otherwise its device discovery fails ("No HIP GPUs are available")
            import torch  # noqa: F401
        except ImportError:
            pass
        _LIB = ctypes.CDLL(p)
        _LIB.gs_last_error.restype = ctypes.c_char_p
        _LIB.gs_version.restype = ctypes.c_char_p
    return _LIB


def _p(a):
    """void* of a numpy array, torch tensor or None."""
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(a.ctypes.data)
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    # torch tensor
    assert a.is_contiguous()
    return ctypes.c_void_p(a.data_ptr())


def _nbytes(a):
    if isinstance(a, np.ndarray):
        return a.nbytes
    return a.numel() * a.element_size()  # torch tensor


def _need(code_name, arrays):
    """Every (name, array, bytes) must hold exactly that many bytes: the C ABI takes bare pointers, so a short proof
    (lengths come from the wire) would otherwise be read past its end.  The reference panics at this point
    (pairing_sum / left_mul asserts, data_structures.rs:495,705): GS_ERR_SHAPE."""
    for name, a, want in arrays:
        if a is None:
            continue
        got = _nbytes(a)
        if got != want:
            raise GsError(1, "%s: %s holds %d bytes, the shape needs %d" % (code_name, name, got, want))


class Engine:
    """One gs_ctx: one GPU, one stream, one CRS."""

    def __init__(self, curve=CURVE_BLS12_381, device=0):
        self.lib = load_library()
        self.curve = curve
        self.ctx = ctypes.c_void_p()
        rc = self.lib.gs_ctx_create(curve, device, ctypes.byref(self.ctx))
        if rc != 0:
            raise GsError(rc, "gs_ctx_create failed (no usable HIP device? there is no CPU fallback)")
        sz = (ctypes.c_size_t * 6)()
        self.lib.gs_sizes(curve, sz)
        self.FQ, self.FR, self.G1, self.G2, self.GT, self.CRS = [int(x) for x in sz]
        self.COM1, self.COM2 = 2 * self.G1, 2 * self.G2

    def close(self):
        if self.ctx:
            # registrations are process-wide and outlive the context: release what THIS engine registered while the
            # arrays are still alive (a stale start -> bytes entry would make a later array that lands on the same
            # addresses pass for page-locked and be moved by DMA against a dead registration)
            for addr in list(getattr(self, "_allocs", {})):
                try:
                    self.lib.gs_host_free(self.ctx, ctypes.c_void_p(addr))
                except Exception:
                    pass
                self._allocs.pop(addr, None)
            for addr in list(getattr(self, "_registered", {})):
                try:
                    self.lib.gs_host_unregister(self.ctx, ctypes.c_void_p(addr))
                except Exception:
                    pass
                self._registered.pop(addr, None)
            self.lib.gs_ctx_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise GsError(rc, (self.lib.gs_last_error(self.ctx) or b"").decode())

    # -- context ------------------------------------------------------------
    def set_stream(self, stream_handle):
        self._chk(self.lib.gs_set_stream(self.ctx, ctypes.c_void_p(stream_handle)))

    def sync(self):
        self._chk(self.lib.gs_sync(self.ctx))

    @staticmethod
    def host_buffer(nbytes):
        """A uint8 array on a fresh anonymous mapping of its own (page-aligned, never part of the allocator's heap):
        what to hand to host_register.  Heap memory is reused without being unmapped, and the runtime may still hold
        its own pin of a freed buffer at the same addresses (include/gs_amd.h, gs_host_register)."""
        import mmap

        return np.frombuffer(mmap.mmap(-1, max(int(nbytes), 1)), dtype=np.uint8)[:nbytes]

    def host_alloc(self, nbytes):
        """A uint8 array over gs_host_alloc memory: pre-faulted, page-locked, registered (DMA-direct in every
        host-pointer call).  Freed by host_free / close."""
        ptr = ctypes.c_void_p()
        self._chk(self.lib.gs_host_alloc(self.ctx, ctypes.c_size_t(max(int(nbytes), 1)), ctypes.byref(ptr)))
        buf = (ctypes.c_uint8 * max(int(nbytes), 1)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=np.uint8)[:nbytes]
        if not hasattr(self, "_allocs"):
            self._allocs = {}
        self._allocs[ptr.value] = arr
        return arr

    def host_free(self, arr):
        addr = arr.ctypes.data
        self._chk(self.lib.gs_host_free(self.ctx, ctypes.c_void_p(addr)))
        getattr(self, "_allocs", {}).pop(addr, None)

    def host_register(self, arr):
        """Page-lock a numpy array the caller reuses across host-pointer calls (gs_host_register): the pipeline then
        moves it by DMA directly, without the staging copy."""
        addr = arr.ctypes.data
        self._chk(self.lib.gs_host_register(self.ctx, ctypes.c_void_p(addr), ctypes.c_size_t(arr.nbytes)))
        # the registration must not outlive the memory: hold the array (for host_buffer() that is the mmap) until
        # host_unregister / close
        if not hasattr(self, "_registered"):
            self._registered = {}
        self._registered[addr] = arr

    def host_unregister(self, arr):
        addr = arr.ctypes.data
        self._chk(self.lib.gs_host_unregister(self.ctx, ctypes.c_void_p(addr)))
        getattr(self, "_registered", {}).pop(addr, None)

    def set_option(self, key, value):
        """Planner override (include/gs_amd.h, gs_set_option): miller_twin, miller_ch, var_tm, var_mo, var_w, red_k, coop_fe, line_tables,
        overlap.  Results never change, only which kernel shapes run."""
        self._chk(self.lib.gs_set_option(self.ctx, key.encode(), int(value)))

    def set_crs(self, crs):
        crs = np.ascontiguousarray(crs).view(np.uint8).reshape(-1)
        assert crs.size == self.CRS, (crs.size, self.CRS)
        self._crs = crs
        self._chk(self.lib.gs_set_crs(self.ctx, _p(crs)))

    def crs_generate(self, p1, p2, scalars, hiding=False):
        """CRS bytes of the reference's shape from generators p1, p2 and scalars (a1, a2, t1, t2); hiding=True gives
        the simulation key of generator.rs:65-77."""
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        out = self._out(self.CRS)
        fn = self.lib.gs_crs_generate_hiding if hiding else self.lib.gs_crs_generate
        p1, p2, scalars = u8(p1), u8(p2), u8(scalars)  # named: the buffers must outlive the call
        self._chk(fn(self.ctx, _p(p1), _p(p2), _p(scalars), _p(out)))
        return out

    # -- shapes ---------------------------------------------------------------
    def shape(self, ty):
        xg, yg = ty in (GS_PPE, GS_MSMEG1), ty in (GS_PPE, GS_MSMEG2)
        return dict(xg=xg, yg=yg, kx=2 if xg else 1, ky=2 if yg else 1, sx=self.G1 if xg else self.FR,
                    sy=self.G2 if yg else self.FR,
                    st={GS_PPE: self.GT, GS_MSMEG1: self.G1, GS_MSMEG2: self.G2, GS_QUAD: self.FR}[ty])

    def _check_prove(self, fn, ty, N, m, n, X, Y, A, B, Gamma, R, S, T, xcoms=None, ycoms=None, pi=None, theta=None):
        if not (0 <= ty <= 3) or m < 1 or n < 1:
            raise GsError(1, "%s: bad equation type or empty variable list" % fn)
        sh = self.shape(ty)
        kx, ky, sx, sy = sh["kx"], sh["ky"], sh["sx"], sh["sy"]
        _need(fn, [("X", X, N * m * sx), ("Y", Y, N * n * sy), ("A", A, N * n * sx), ("B", B, N * m * sy),
                   ("Gamma", Gamma, N * m * n * self.FR), ("R", R, N * m * kx * self.FR),
                   ("S", S, N * n * ky * self.FR), ("T", T, N * ky * kx * self.FR),
                   ("xcoms", xcoms, N * m * self.COM1), ("ycoms", ycoms, N * n * self.COM2),
                   ("pi", pi, N * kx * self.COM2), ("theta", theta, N * ky * self.COM1)])

    def _check_verify(self, fn, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta):
        if not (0 <= ty <= 3) or m < 1 or n < 1:
            raise GsError(1, "%s: bad equation type or empty variable list" % fn)
        sh = self.shape(ty)
        kx, ky, sx, sy, st = sh["kx"], sh["ky"], sh["sx"], sh["sy"], sh["st"]
        _need(fn, [("A", A, N * n * sx), ("B", B, N * m * sy), ("Gamma", Gamma, N * m * n * self.FR),
                   ("target", target, N * st), ("xcoms", xcoms, N * m * self.COM1),
                   ("ycoms", ycoms, N * n * self.COM2), ("pi", pi, N * kx * self.COM2),
                   ("theta", theta, N * ky * self.COM1)])

    # -- host entry points (numpy uint8/uint64 arrays) --------------------------
    def _out(self, nbytes):
        return np.zeros(nbytes, dtype=np.uint8)

    def commit(self, kind, vars_, rand):
        fn = {"g1": self.lib.gs_commit_g1, "g2": self.lib.gs_commit_g2, "fr_b1": self.lib.gs_commit_fr_b1,
              "fr_b2": self.lib.gs_commit_fr_b2}[kind]
        vsz = {"g1": self.G1, "g2": self.G2, "fr_b1": self.FR, "fr_b2": self.FR}[kind]
        osz = self.COM1 if kind in ("g1", "fr_b1") else self.COM2
        v = np.ascontiguousarray(vars_).view(np.uint8).reshape(-1)
        r = np.ascontiguousarray(rand).view(np.uint8).reshape(-1)
        n = v.size // vsz
        out = self._out(n * osz)
        self._chk(fn(self.ctx, ctypes.c_size_t(n), _p(v), _p(r), _p(out)))
        return out.reshape(n, osz)

    def prove_batch(self, ty, N, m, n, X, Y, A, B, Gamma, R, S, T, want_coms=True, out=None):
        """`out`: dict of preallocated uint8 arrays xcoms, ycoms, pi, theta to write into (e.g. page-locked ones)."""
        sh = self.shape(ty)
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        X, Y, A, B, Gamma, R, S, T = map(u8, (X, Y, A, B, Gamma, R, S, T))
        self._check_prove("gs_prove_batch", ty, N, m, n, X, Y, A, B, Gamma, R, S, T)
        if out is not None:
            xc, yc, pi, th = (out.get(k) for k in ("xcoms", "ycoms", "pi", "theta"))
            _need("gs_prove_batch", [("xcoms", xc, N * m * self.COM1), ("ycoms", yc, N * n * self.COM2),
                                     ("pi", pi, N * sh["kx"] * self.COM2), ("theta", th, N * sh["ky"] * self.COM1)])
            if pi is None or th is None:
                raise GsError(3, "gs_prove_batch: out needs pi and theta")
        else:
            xc = self._out(N * m * self.COM1) if want_coms else None
            yc = self._out(N * n * self.COM2) if want_coms else None
            pi = self._out(N * sh["kx"] * self.COM2)
            th = self._out(N * sh["ky"] * self.COM1)
        self._chk(self.lib.gs_prove_batch(self.ctx, ty, ctypes.c_size_t(N), m, n, _p(u8(X)), _p(u8(Y)), _p(u8(A)),
                                          _p(u8(B)), _p(u8(Gamma)), _p(u8(R)), _p(u8(S)), _p(u8(T)), _p(xc), _p(yc),
                                          _p(pi), _p(th)))
        return dict(xcoms=xc, ycoms=yc, pi=pi, theta=th)

    def verify_batch(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta, ok=None):
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        if ok is None:
            ok = np.zeros(N, dtype=np.uint8)
        _need("gs_verify_batch", [("ok", ok, N)])
        A, B, Gamma, target, xcoms, ycoms, pi, theta = map(u8, (A, B, Gamma, target, xcoms, ycoms, pi, theta))
        self._check_verify("gs_verify_batch", ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta)
        self._chk(self.lib.gs_verify_batch(self.ctx, ty, ctypes.c_size_t(N), m, n, _p(u8(A)), _p(u8(B)),
                                           _p(u8(Gamma)), _p(u8(target)), _p(u8(xcoms)), _p(u8(ycoms)), _p(u8(pi)),
                                           _p(u8(theta)), _p(ok)))
        return ok

    def prove_statement(self, ty, E, m, n, X, Y, A, B, Gamma, R, S, T, want_coms=True):
        """E equations of one type over the SAME variables X[m], Y[n] (commit randomness R, S): commitments once,
        one proof per equation (gs_prove_statement)."""
        sh = self.shape(ty)
        kx, ky, sx, sy = sh["kx"], sh["ky"], sh["sx"], sh["sy"]
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        X, Y, A, B, Gamma, R, S, T = map(u8, (X, Y, A, B, Gamma, R, S, T))
        if not (0 <= ty <= 3) or m < 1 or n < 1:
            raise GsError(1, "gs_prove_statement: bad equation type or empty variable list")
        _need("gs_prove_statement", [("X", X, m * sx), ("Y", Y, n * sy), ("A", A, E * n * sx), ("B", B, E * m * sy),
                                     ("Gamma", Gamma, E * m * n * self.FR), ("R", R, m * kx * self.FR),
                                     ("S", S, n * ky * self.FR), ("T", T, E * ky * kx * self.FR)])
        xc = self._out(m * self.COM1) if want_coms else None
        yc = self._out(n * self.COM2) if want_coms else None
        pi, th = self._out(E * kx * self.COM2), self._out(E * ky * self.COM1)
        self._chk(self.lib.gs_prove_statement(self.ctx, ty, ctypes.c_size_t(E), m, n, _p(X), _p(Y), _p(A), _p(B),
                                              _p(Gamma), _p(R), _p(S), _p(T), _p(xc), _p(yc), _p(pi), _p(th)))
        return dict(xcoms=xc, ycoms=yc, pi=pi, theta=th)

    def verify_statement(self, ty, E, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta):
        sh = self.shape(ty)
        kx, ky, sx, sy, st = sh["kx"], sh["ky"], sh["sx"], sh["sy"], sh["st"]
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        A, B, Gamma, target, xcoms, ycoms, pi, theta = map(u8, (A, B, Gamma, target, xcoms, ycoms, pi, theta))
        if not (0 <= ty <= 3) or m < 1 or n < 1:
            raise GsError(1, "gs_verify_statement: bad equation type or empty variable list")
        _need("gs_verify_statement", [("A", A, E * n * sx), ("B", B, E * m * sy), ("Gamma", Gamma, E * m * n * self.FR),
                                      ("target", target, E * st), ("xcoms", xcoms, m * self.COM1),
                                      ("ycoms", ycoms, n * self.COM2), ("pi", pi, E * kx * self.COM2),
                                      ("theta", theta, E * ky * self.COM1)])
        ok = np.zeros(E, dtype=np.uint8)
        self._chk(self.lib.gs_verify_statement(self.ctx, ty, ctypes.c_size_t(E), m, n, _p(A), _p(B), _p(Gamma),
                                               _p(target), _p(xcoms), _p(ycoms), _p(pi), _p(theta), _p(ok)))
        return ok

    def verify_batch_rlc(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta, rho):
        """Batched verifier (host buffers).  rho: uint64[N*4].  Returns (ok_all, acc_pair_bytes)."""
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        rho = np.ascontiguousarray(rho, dtype=np.uint64).reshape(-1)
        A, B, Gamma, target, xcoms, ycoms, pi, theta = map(u8, (A, B, Gamma, target, xcoms, ycoms, pi, theta))
        self._check_verify("gs_verify_batch_rlc", ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta)
        _need("gs_verify_batch_rlc", [("rho", rho, 32 * N)])
        acc = self._out(2 * self.GT)
        ok = np.zeros(1, dtype=np.uint8)
        self._chk(self.lib.gs_verify_batch_rlc(self.ctx, ty, ctypes.c_size_t(N), m, n, _p(u8(A)), _p(u8(B)),
                                               _p(u8(Gamma)), _p(u8(target)), _p(u8(xcoms)), _p(u8(ycoms)),
                                               _p(u8(pi)), _p(u8(theta)), _p(rho), _p(acc), _p(ok)))
        return int(ok[0]), acc

    def gt_finalize_dev(self, accs_dev, count):
        """the same with `count` pairs in device memory (torch tensor)."""
        _need("gs_gt_finalize_dev", [("accs", accs_dev, count * 2 * self.GT)])
        ok = np.zeros(1, dtype=np.uint8)
        self._chk(self.lib.gs_gt_finalize_dev(self.ctx, ctypes.c_size_t(count), _p(accs_dev), _p(ok)))
        return int(ok[0])

    def gt_finalize(self, accs):
        """accs: bytes of `count` accumulator pairs (host).  FE(prod acc[i][0]) == prod acc[i][1] ?"""
        a = np.ascontiguousarray(accs).view(np.uint8).reshape(-1)
        count = a.size // (2 * self.GT)
        ok = np.zeros(1, dtype=np.uint8)
        self._chk(self.lib.gs_gt_finalize(self.ctx, ctypes.c_size_t(count), _p(a), _p(ok)))
        return int(ok[0])

    def mat_left_mul(self, group, rows, k, lhs, col):
        fn = self.lib.gs_mat_left_mul_com1 if group == 1 else self.lib.gs_mat_left_mul_com2
        osz = self.COM1 if group == 1 else self.COM2
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        out = self._out(rows * osz)
        lhs, col = u8(lhs), u8(col)
        _need("gs_mat_left_mul", [("lhs", lhs, rows * k * self.FR), ("col", col, k * osz)])
        self._chk(fn(self.ctx, rows, k, _p(lhs), _p(col), _p(out)))
        return out.reshape(rows, osz)

    def fr_matmul(self, lhs, rhs):
        """Matrix<Fr> product (lists of rows of 4 x u64 Montgomery scalars) on the prover's preparation kernels."""
        rows, inner, cols = len(lhs), len(lhs[0]), len(rhs[0])
        assert len(rhs) == inner and all(len(r) == inner for r in lhs) and all(len(r) == cols for r in rhs)
        u8 = lambda mat: np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.uint64).reshape(-1) for r in mat
                                                        for v in r])).view(np.uint8).reshape(-1)
        out = self._out(rows * cols * self.FR)
        a, b = u8(lhs), u8(rhs)  # named: the buffers must outlive the call
        self._chk(self.lib.gs_fr_matmul(self.ctx, rows, inner, cols, _p(a), _p(b), _p(out)))
        o = out.view(np.uint64).reshape(rows, cols, self.FR // 8)
        return [[o[i, j].copy() for j in range(cols)] for i in range(rows)]

    def pairing_sum(self, k, x, y):
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        out = self._out(4 * self.GT)
        x, y = u8(x), u8(y)
        _need("gs_pairing_sum", [("x", x, k * self.COM1), ("y", y, k * self.COM2)])
        self._chk(self.lib.gs_pairing_sum(self.ctx, k, _p(x), _p(y), _p(out)))
        return out.reshape(4, self.GT)

    def g_mul_batch(self, group, points, scalars, broadcast=False):
        fn = self.lib.gs_g1_mul_batch if group == 1 else self.lib.gs_g2_mul_batch
        gsz = self.G1 if group == 1 else self.G2
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        k = u8(scalars)
        n = k.size // self.FR
        out = self._out(n * gsz)
        points = u8(points)
        _need("gs_g_mul_batch", [("points", points, (1 if broadcast else n) * gsz)])
        self._chk(fn(self.ctx, ctypes.c_size_t(n), _p(points), 1 if broadcast else 0, _p(k), _p(out)))
        return out.reshape(n, gsz)

    def multi_pairing_batch(self, n, k, P, Q):
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        out = self._out(n * self.GT)
        P, Q = u8(P), u8(Q)
        _need("gs_multi_pairing_batch", [("P", P, n * k * self.G1), ("Q", Q, n * k * self.G2)])
        self._chk(self.lib.gs_multi_pairing_batch(self.ctx, ctypes.c_size_t(n), k, _p(P), _p(Q), _p(out)))
        return out.reshape(n, self.GT)

    # -- device entry points (torch CUDA uint8 tensors; asynchronous) --------------
    def prove_batch_dev(self, ty, N, m, n, X, Y, A, B, Gamma, R, S, T, xcoms, ycoms, pi, theta):
        self._check_prove("gs_prove_batch_dev", ty, N, m, n, X, Y, A, B, Gamma, R, S, T, xcoms, ycoms, pi, theta)
        self._chk(self.lib.gs_prove_batch_dev(self.ctx, ty, ctypes.c_size_t(N), m, n, _p(X), _p(Y), _p(A), _p(B),
                                              _p(Gamma), _p(R), _p(S), _p(T), _p(xcoms), _p(ycoms), _p(pi),
                                              _p(theta)))

    def verify_batch_dev(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta, ok):
        self._check_verify("gs_verify_batch_dev", ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta)
        _need("gs_verify_batch_dev", [("ok", ok, N)])
        self._chk(self.lib.gs_verify_batch_dev(self.ctx, ty, ctypes.c_size_t(N), m, n, _p(A), _p(B), _p(Gamma),
                                               _p(target), _p(xcoms), _p(ycoms), _p(pi), _p(theta), _p(ok)))

    def verify_batch_rlc_dev(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta, rho, acc):
        self._check_verify("gs_verify_batch_rlc_dev", ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta)
        _need("gs_verify_batch_rlc_dev", [("rho", rho, 32 * N), ("acc", acc, 2 * self.GT)])
        self._chk(self.lib.gs_verify_batch_rlc_dev(self.ctx, ty, ctypes.c_size_t(N), m, n, _p(A), _p(B), _p(Gamma),
                                                   _p(target), _p(xcoms), _p(ycoms), _p(pi), _p(theta), _p(rho),
                                                   _p(acc)))

    def g_mul_batch_dev(self, group, n, points, broadcast, scalars, out):
        fn = self.lib.gs_g1_mul_batch_dev if group == 1 else self.lib.gs_g2_mul_batch_dev
        self._chk(fn(self.ctx, ctypes.c_size_t(n), _p(points), 1 if broadcast else 0, _p(scalars), _p(out)))

    def multi_pairing_batch_dev(self, n, k, P, Q, out):
        self._chk(self.lib.gs_multi_pairing_batch_dev(self.ctx, ctypes.c_size_t(n), k, _p(P), _p(Q), _p(out)))

    def gt_pow_batch_dev(self, n, base, k, out):
        self._chk(self.lib.gs_gt_pow_batch_dev(self.ctx, ctypes.c_size_t(n), _p(base), _p(k), _p(out)))

    # -- mixed batches / mixed-type Statements (gs_prove_mixed, gs_verify_mixed) -----------
    def _part_sizes(self, ty, N, m, n, shared):
        sh = self.shape(ty)
        kx, ky, sx, sy, st = sh["kx"], sh["ky"], sh["sx"], sh["sy"], sh["st"]
        V = 1 if shared else N
        return dict(X=V * m * sx, Y=V * n * sy, A=N * n * sx, B=N * m * sy, Gamma=N * m * n * self.FR,
                    R=V * m * kx * self.FR, S=V * n * ky * self.FR, T=N * ky * kx * self.FR, target=N * st,
                    xcoms=V * m * self.COM1, ycoms=V * n * self.COM2, pi=N * kx * self.COM2, theta=N * ky * self.COM1,
                    ok=N)

    def _parts(self, fn, parts, struct, in_keys, out_keys, dev):
        """parts: dicts with ty, N, m, n, the input arrays and (dev) the output arrays; host outputs are allocated
        here (xcoms / ycoms only when the part asks with want_coms, default True).  Every length is checked before a
        pointer crosses the ABI."""
        if len(parts) > GS_MIXED_MAX:
            raise GsError(3, "%s: at most %d parts" % (fn, GS_MIXED_MAX))
        arr = (struct * max(len(parts), 1))()
        keep, outs = [], []
        for i, p in enumerate(parts):
            ty, N, m, n, shared = p["ty"], p["N"], p["m"], p["n"], bool(p.get("shared", False))
            if not (0 <= ty <= 3) or m < 1 or n < 1:
                raise GsError(1, "%s: bad equation type or empty variable list" % fn)
            want = self._part_sizes(ty, N, m, n, shared)
            a = arr[i]
            a.equ_type, a.N, a.m, a.n, a.shared_vars = ty, N, m, n, 1 if shared else 0
            o = {}
            for k in in_keys:
                v = p[k]
                if not dev:
                    v = np.ascontiguousarray(v).view(np.uint8).reshape(-1)
                _need(fn, [(k, v, want[k])])
                keep.append(v)
                setattr(a, k, _p(v).value)
            for k in out_keys:
                if dev:
                    v = p.get(k)
                    if v is not None:
                        _need(fn, [(k, v, want[k])])
                elif k in ("xcoms", "ycoms") and not p.get("want_coms", True):
                    v = None
                else:
                    v = np.zeros(want[k], dtype=np.uint8)
                keep.append(v)
                o[k] = v
                setattr(a, k, _p(v).value)
            outs.append(o)
        return arr, keep, outs

    def prove_mixed(self, parts):
        """Several sub-batches (any types / shapes) in one call, host arrays.  -> list of dicts of outputs."""
        arr, keep, outs = self._parts("gs_prove_mixed", parts, ProvePart, ("X", "Y", "A", "B", "Gamma", "R", "S", "T"),
                                      ("xcoms", "ycoms", "pi", "theta"), False)
        self._chk(self.lib.gs_prove_mixed(self.ctx, len(parts), arr))
        return outs

    def verify_mixed(self, parts):
        arr, keep, outs = self._parts("gs_verify_mixed", parts, VerifyPart,
                                      ("A", "B", "Gamma", "target", "xcoms", "ycoms", "pi", "theta"), ("ok",), False)
        self._chk(self.lib.gs_verify_mixed(self.ctx, len(parts), arr))
        return [o["ok"] for o in outs]

    def prove_mixed_dev(self, parts):
        """Device tensors in the part dicts (outputs included); asynchronous on the context's stream."""
        arr, keep, _ = self._parts("gs_prove_mixed_dev", parts, ProvePart, ("X", "Y", "A", "B", "Gamma", "R", "S", "T"),
                                   ("xcoms", "ycoms", "pi", "theta"), True)
        self._chk(self.lib.gs_prove_mixed_dev(self.ctx, len(parts), arr))

    def verify_mixed_dev(self, parts):
        arr, keep, _ = self._parts("gs_verify_mixed_dev", parts, VerifyPart,
                                   ("A", "B", "Gamma", "target", "xcoms", "ycoms", "pi", "theta"), ("ok",), True)
        self._chk(self.lib.gs_verify_mixed_dev(self.ctx, len(parts), arr))

    # -- wire format (ark-serialize byte strings <-> boundary arrays) -------------------
    def wire_sizes(self):
        out = (ctypes.c_size_t * 6)()
        self._chk(self.lib.gs_wire_sizes(self.curve, out))
        return dict(zip(("g1c", "g1u", "g2c", "g2u", "fr", "gt"), [int(v) for v in out]))

    def wire_encode(self, kind, arr, compressed=True):
        """kind in g1|g2|fr|gt; arr: (n, element bytes) boundary values -> (n, wire bytes) uint8."""
        ws = self.wire_sizes()
        arr = np.ascontiguousarray(arr).view(np.uint8)
        esz = {"g1": self.G1, "g2": self.G2, "fr": self.FR, "gt": self.GT}[kind]
        n = arr.size // esz
        wsz = ws[kind + ("c" if compressed else "u")] if kind in ("g1", "g2") else ws[kind]
        out = np.zeros((n, wsz), dtype=np.uint8)
        if n == 0:
            return out
        if kind in ("g1", "g2"):
            fn = self.lib.gs_wire_encode_g1 if kind == "g1" else self.lib.gs_wire_encode_g2
            self._chk(fn(self.ctx, ctypes.c_size_t(n), 1 if compressed else 0, _p(arr), _p(out)))
        else:
            fn = self.lib.gs_wire_encode_fr if kind == "fr" else self.lib.gs_wire_encode_gt
            self._chk(fn(self.ctx, ctypes.c_size_t(n), _p(arr), _p(out)))
        return out

    def wire_decode(self, kind, buf, compressed=True, validate=True):
        """-> (values (n, element bytes) uint8, ok (n,) uint8)."""
        ws = self.wire_sizes()
        buf = np.ascontiguousarray(buf).view(np.uint8)
        esz = {"g1": self.G1, "g2": self.G2, "fr": self.FR, "gt": self.GT}[kind]
        wsz = ws[kind + ("c" if compressed else "u")] if kind in ("g1", "g2") else ws[kind]
        n = buf.size // wsz
        out = np.zeros((n, esz), dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        if n == 0:
            return out, ok
        if kind in ("g1", "g2"):
            fn = self.lib.gs_wire_decode_g1 if kind == "g1" else self.lib.gs_wire_decode_g2
            self._chk(fn(self.ctx, ctypes.c_size_t(n), 1 if compressed else 0, 1 if validate else 0, _p(buf), _p(out),
                         _p(ok)))
        elif kind == "fr":
            self._chk(self.lib.gs_wire_decode_fr(self.ctx, ctypes.c_size_t(n), _p(buf), _p(out), _p(ok)))
        else:
            self._chk(self.lib.gs_wire_decode_gt(self.ctx, ctypes.c_size_t(n), 1 if validate else 0, _p(buf), _p(out),
                                                 _p(ok)))
        return out, ok

    def validate_points(self, group, pts):
        """ok[i] = 1 iff point i (boundary limbs; group 1 = G1, 2 = G2) is canonical, on the curve (or the identity)
        and in the prime-order subgroup (gs_validate_points)."""
        pts = np.ascontiguousarray(pts).view(np.uint8).reshape(-1)
        per = self.G1 if group == 1 else self.G2
        assert group in (1, 2) and pts.size % per == 0
        n = pts.size // per
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self.lib.gs_validate_points(self.ctx, group, ctypes.c_size_t(n), _p(pts), _p(ok)))
        return ok

    # -- profiling hook ---------------------------------------------------------------
    def prof_enable(self, on=True):
        self.lib.gs_prof_enable(self.ctx, 1 if on else 0)

    def prof_reset(self):
        self.lib.gs_prof_reset(self.ctx)

    def prof_get(self):
        out = []
        i = 0
        while True:
            name = ctypes.create_string_buffer(128)
            ms = ctypes.c_double()
            n = ctypes.c_uint64()
            if self.lib.gs_prof_get(self.ctx, i, name, ctypes.c_size_t(128), ctypes.byref(ms), ctypes.byref(n)) != 0:
                break
            out.append((name.value.decode(), ms.value, n.value))
            i += 1
        return out

    def prof_get_work(self):
        """{kernel name: (lanes, work items)} for the kernels profiled since prof_reset()."""
        out = {}
        for i, (name, _, _) in enumerate(self.prof_get()):
            lanes, work = ctypes.c_uint64(), ctypes.c_uint64()
            self._chk(self.lib.gs_prof_get_work(self.ctx, i, ctypes.byref(lanes), ctypes.byref(work)))
            out[name] = (lanes.value, work.value)
        return out

    def prof_get_clock(self):
        """{kernel name: GHz} -- the clock the profiled launches ran at, stamped inside them (s_memtime /
        s_memrealtime per wave, gs_prof_get_clock); absent for kernels that are not segmented launches."""
        out = {}
        for i, (name, _, _) in enumerate(self.prof_get()):
            ghz = ctypes.c_double()
            self._chk(self.lib.gs_prof_get_clock(self.ctx, i, ctypes.byref(ghz)))
            if ghz.value > 0:
                out[name] = ghz.value
        return out


class MultiEngine:
    """gs_ctx_create_multi: the devices of one node behind one handle.  Host arrays in, host arrays out; the batch is
    cut into contiguous equation blocks, one per device (include/gs_amd.h, multi-GPU section)."""

    def __init__(self, curve=CURVE_BLS12_381, devices=(0,), shared_devices=False):
        """shared_devices=True (GS_MULTI_SHARED_DEVICES) lets several shards name the same ordinal: the ndev > 1 split
        on a one-GPU box."""
        self.lib = load_library()
        self.lib.gs_multi_last_error.restype = ctypes.c_char_p
        self.lib.gs_multi_exchange_note.restype = ctypes.c_char_p
        self.lib.gs_multi_ctx.restype = ctypes.c_void_p
        self.curve = curve
        self.devices = list(devices)
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        self.h = ctypes.c_void_p()
        rc = self.lib.gs_ctx_create_multi_ex(curve, arr, len(self.devices),
                                             GS_MULTI_SHARED_DEVICES if shared_devices else 0, ctypes.byref(self.h))
        if rc != 0:
            raise GsError(rc, "gs_ctx_create_multi failed (no usable HIP device? there is no CPU fallback)")
        sz = (ctypes.c_size_t * 6)()
        self.lib.gs_sizes(curve, sz)
        self.FQ, self.FR, self.G1, self.G2, self.GT, self.CRS = [int(x) for x in sz]
        self.COM1, self.COM2 = 2 * self.G1, 2 * self.G2

    shape = Engine.shape
    _check_prove = Engine._check_prove
    _check_verify = Engine._check_verify

    def close(self):
        if self.h:
            self.lib.gs_multi_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise GsError(rc, (self.lib.gs_multi_last_error(self.h) or b"").decode())

    def set_option(self, key, value):
        if key.startswith("exchange_"):  # options of the multi-GPU layer itself (gs_multi_set_option)
            return self._chk(self.lib.gs_multi_set_option(self.h, key.encode(), int(value)))
        for i in range(len(self.devices)):
            ctx = ctypes.c_void_p(self.lib.gs_multi_ctx(self.h, i))
            if self.lib.gs_set_option(ctx, key.encode(), int(value)) != 0:
                raise GsError(3, "gs_set_option(%s)" % key)

    def shard(self, N, i):
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        self._chk(self.lib.gs_multi_shard(self.h, ctypes.c_size_t(N), i, ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    def uses_rccl(self):
        return bool(self.lib.gs_multi_uses_rccl(self.h))

    def exchange_note(self):
        return (self.lib.gs_multi_exchange_note(self.h) or b"").decode()

    def sync(self):
        self._chk(self.lib.gs_multi_sync(self.h))

    # -- device-resident shards: lists of ndev torch tensors (entry i on devices[i], shard i's block) -----------------
    def _pp(self, lst):
        if lst is None:
            return None
        assert len(lst) == len(self.devices)
        return (ctypes.c_void_p * len(lst))(*[_p(t).value for t in lst])

    def prove_batch_dev(self, ty, N, m, n, X, Y, A, B, Gamma, R, S, T, xcoms, ycoms, pi, theta):
        for i in range(len(self.devices)):
            lo, hi = self.shard(N, i)
            pick = lambda l: None if l is None else l[i]
            if hi > lo:
                self._check_prove("gs_multi_prove_batch_dev", ty, hi - lo, m, n, X[i], Y[i], A[i], B[i], Gamma[i], R[i],
                                  S[i], T[i], pick(xcoms), pick(ycoms), pi[i], theta[i])
        args = [self._pp(a) for a in (X, Y, A, B, Gamma, R, S, T, xcoms, ycoms, pi, theta)]
        self._chk(self.lib.gs_multi_prove_batch_dev(self.h, ty, ctypes.c_size_t(N), m, n, *args))

    def verify_batch_dev(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta, ok):
        for i in range(len(self.devices)):
            lo, hi = self.shard(N, i)
            if hi > lo:
                self._check_verify("gs_multi_verify_batch_dev", ty, hi - lo, m, n, A[i], B[i], Gamma[i], target[i],
                                   xcoms[i], ycoms[i], pi[i], theta[i])
                _need("gs_multi_verify_batch_dev", [("ok", ok[i], hi - lo)])
        args = [self._pp(a) for a in (A, B, Gamma, target, xcoms, ycoms, pi, theta, ok)]
        self._chk(self.lib.gs_multi_verify_batch_dev(self.h, ty, ctypes.c_size_t(N), m, n, *args))

    def verify_batch_rlc_dev(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta, rho):
        """rho: list of per-shard uint64 device tensors (4 per equation of the shard's block).  -> (ok_all, pairs)"""
        for i in range(len(self.devices)):
            lo, hi = self.shard(N, i)
            if hi > lo:
                self._check_verify("gs_multi_verify_batch_rlc_dev", ty, hi - lo, m, n, A[i], B[i], Gamma[i], target[i],
                                   xcoms[i], ycoms[i], pi[i], theta[i])
                _need("gs_multi_verify_batch_rlc_dev", [("rho", rho[i], 32 * (hi - lo))])
        args = [self._pp(a) for a in (A, B, Gamma, target, xcoms, ycoms, pi, theta, rho)]
        acc = np.zeros(len(self.devices) * 2 * self.GT, dtype=np.uint8)
        okb = np.zeros(1, dtype=np.uint8)
        self._chk(self.lib.gs_multi_verify_batch_rlc_dev(self.h, ty, ctypes.c_size_t(N), m, n, *args, _p(acc), _p(okb)))
        return int(okb[0]), acc

    def set_crs(self, crs):
        crs = np.ascontiguousarray(crs).view(np.uint8).reshape(-1)
        assert crs.size == self.CRS
        self._chk(self.lib.gs_multi_set_crs(self.h, _p(crs)))

    def prove_batch(self, ty, N, m, n, X, Y, A, B, Gamma, R, S, T, want_coms=True):
        sh = self.shape(ty)
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        X, Y, A, B, Gamma, R, S, T = map(u8, (X, Y, A, B, Gamma, R, S, T))
        self._check_prove("gs_multi_prove_batch", ty, N, m, n, X, Y, A, B, Gamma, R, S, T)
        z = lambda k: np.zeros(k, dtype=np.uint8)
        xc = z(N * m * self.COM1) if want_coms else None
        yc = z(N * n * self.COM2) if want_coms else None
        pi, th = z(N * sh["kx"] * self.COM2), z(N * sh["ky"] * self.COM1)
        self._chk(self.lib.gs_multi_prove_batch(self.h, ty, ctypes.c_size_t(N), m, n, _p(X), _p(Y), _p(A), _p(B),
                                                _p(Gamma), _p(R), _p(S), _p(T), _p(xc), _p(yc), _p(pi), _p(th)))
        return dict(xcoms=xc, ycoms=yc, pi=pi, theta=th)

    def verify_batch(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta):
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        A, B, Gamma, target, xcoms, ycoms, pi, theta = map(u8, (A, B, Gamma, target, xcoms, ycoms, pi, theta))
        self._check_verify("gs_multi_verify_batch", ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta)
        ok = np.zeros(N, dtype=np.uint8)
        self._chk(self.lib.gs_multi_verify_batch(self.h, ty, ctypes.c_size_t(N), m, n, _p(A), _p(B), _p(Gamma),
                                                 _p(target), _p(xcoms), _p(ycoms), _p(pi), _p(theta), _p(ok)))
        return ok

    def verify_batch_rlc(self, ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta, rho):
        """-> (ok_all, the ndev gathered accumulator pairs as bytes)"""
        u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        A, B, Gamma, target, xcoms, ycoms, pi, theta = map(u8, (A, B, Gamma, target, xcoms, ycoms, pi, theta))
        self._check_verify("gs_multi_verify_batch_rlc", ty, N, m, n, A, B, Gamma, target, xcoms, ycoms, pi, theta)
        rho = np.ascontiguousarray(rho, dtype=np.uint64).reshape(-1)
        _need("gs_multi_verify_batch_rlc", [("rho", rho, 32 * N)])
        acc = np.zeros(len(self.devices) * 2 * self.GT, dtype=np.uint8)
        ok = np.zeros(1, dtype=np.uint8)
        self._chk(self.lib.gs_multi_verify_batch_rlc(self.h, ty, ctypes.c_size_t(N), m, n, _p(A), _p(B), _p(Gamma),
                                                     _p(target), _p(xcoms), _p(ycoms), _p(pi), _p(theta), _p(rho),
                                                     _p(acc), _p(ok)))
        return int(ok[0]), acc
