"""ctypes binding of include/nbody3d_hip.h.

Mirrors the reference's host-side protocol for the hot path (file:line relative
to /root/reference):

    Simulation(n).init(bodies, vel)   nbody3d.js:177-199  (create + writeBuffer)
    sim.step(dt)                      nbody3d.js:470,474-480,489-490
    sim.simulate(k, dt)               k back-to-back frames of the above
    sim.read()                        util.js:163-178     (exportSimulation copies)
    sim.restore(bodies, vel, accel)   util.js:230-244     (importSimulation)
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# NB_ENGINE_LIB: load another build of the same library (the -DNB_TUNING calibration build of csrc/Makefile)
_LIB_PATH = os.environ.get("NB_ENGINE_LIB") or os.path.normpath(os.path.join(_PKG, "..", "csrc", "libnbody3d_hip.so"))

NB_F32, NB_F64 = 0, 1
NB_FLAG_EXT_STREAM = 1
NB_FLAG_LDS_ONLY = 4
NB_FLAG_NO_FUSE = 8
NB_FLAG_POISON = 16
NB_FLAG_JPK_FENCED = 32
NB_FLAG_NO_SYM = 64
NB_FLAG_SYM_SHARD = 128
NB_FLAG_WHOLE_SWEEPS = 256
NB_RCCL_ID_BYTES = 128
NB_RCCL_OVERLAP = 1
NB_MULTI_PEER, NB_MULTI_RCCL, NB_MULTI_PEER_OVERLAP = 0, 1, 2
NB_NOT_READY = 7
STATUS = {0: "NB_OK", 1: "NB_ERR_INVALID", 2: "NB_ERR_NO_DEVICE", 3: "NB_ERR_HIP",
          4: "NB_ERR_STATE", 5: "NB_ERR_NOMEM", 6: "NB_ERR_COMM", 7: "NB_NOT_READY"}
ABI_VERSION = 2      # NB_ABI_VERSION (major)
ABI_MINOR = 3        # NB_ABI_MINOR this binding was written against


class NBodyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS.get(code, code), msg))
        self.code = code


class nb_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n", C.c_uint32), ("precision", C.c_uint32), ("tile", C.c_uint32),
        ("eps2", C.c_double), ("device", C.c_int32), ("shard_begin", C.c_uint32), ("shard_count", C.c_uint32),
        ("ext_stream", C.c_void_p), ("ext_bodies", C.c_void_p),
        ("force_variant", C.c_uint32), ("jsplit", C.c_uint32), ("flags", C.c_uint32), ("layer_budget_mib", C.c_uint32),
        ("reserved", C.c_uint32 * 4),
    ]


class nb_plan_info(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + [(k, C.c_uint32) for k in (
        "kind", "ipl", "ls", "x", "jsplit", "j_per_split", "own_split0", "own_splits",
        "sym", "symw", "sym_rank", "sym_np", "sym_layers", "sym_g0", "sym_g1")] + [
        ("sym_plan", C.c_uint32 * 11), ("tab_len", C.c_uint32), ("variant", C.c_char * 112),
        ("sym_ups", C.c_uint32), ("sym_spill_rows", C.c_uint32), ("sym_rank_plan", C.c_uint32 * 16),
        ("sym_pass", C.c_uint32), ("sym_passes", C.c_uint32), ("sym_pass_k_lo", C.c_uint32), ("sym_pass_k_hi", C.c_uint32),
        ("sym_pass_d0", C.c_uint32), ("sym_local", C.c_uint32)]


class nb_step_timing(C.Structure):          # include/nbody3d_hip.h
    _fields_ = [("struct_size", C.c_uint32), ("launches", C.c_uint32), ("force_ms", C.c_double), ("sym_reduce_ms", C.c_double),
                ("reduce_scatter_ms", C.c_double), ("integrate_ms", C.c_double), ("allgather_ms", C.c_double),
                ("span_ms", C.c_double), ("reduce_scatters", C.c_uint32), ("allgathers", C.c_uint32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p)
EXCHANGE_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

# every symbol include/nbody3d_hip.h and include/nbody3d_hip_plan.h declare (tests check the export list)
SYMBOLS = ["nb_abi_version", "nb_device_count", "nb_create", "nb_destroy", "nb_upload", "nb_set_params", "nb_step",
           "nb_download", "nb_sync", "nb_last_error", "nb_device_ptr", "nb_set_exchange",
           "nb_set_exchange_overlapped", "nb_enable_timing",
           "nb_kernel_times", "nb_variant_name", "nb_diagnostics",
           "nb_multi_create", "nb_multi_destroy", "nb_multi_upload", "nb_multi_set_params", "nb_multi_step",
           "nb_multi_download", "nb_multi_sync", "nb_multi_last_error", "nb_multi_variant_name",
           "nb_multi_diagnostics", "nb_multi_set_collective", "nb_multi_collective_info",
           "nb_rccl_unique_id", "nb_rccl_attach", "nb_rccl_detach", "nb_rccl_info",
           "nb_step_times", "nb_step_times2", "nb_integrate_pass", "nb_force_pass", "nb_frame_request", "nb_frame_acquire", "nb_shape_info", "nb_plan_query",
           "nb_abi_minor"]

_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Loads libnbody3d_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise NBodyError(3, "HIP engine not built: %s is missing (run __graft_entry__.build() or "
                            "make -C nbody3d-webgpu_amd/csrc)" % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.nb_abi_version.restype = C.c_uint32
    L.nb_abi_minor.restype = C.c_uint32
    L.nb_device_count.restype = C.c_int
    L.nb_create.argtypes = [C.POINTER(nb_config), C.POINTER(vp)]
    L.nb_destroy.argtypes = [vp]
    L.nb_destroy.restype = None
    L.nb_upload.argtypes = [vp, vp, vp, vp]
    L.nb_set_params.argtypes = [vp, C.c_double, C.c_double]
    L.nb_step.argtypes = [vp, C.c_uint32]
    L.nb_download.argtypes = [vp, vp, vp, vp]
    L.nb_sync.argtypes = [vp]
    L.nb_last_error.argtypes = [vp]
    L.nb_last_error.restype = C.c_char_p
    L.nb_device_ptr.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.nb_set_exchange.argtypes = [vp, EXCHANGE_FN, vp]
    L.nb_set_exchange_overlapped.argtypes = [vp, EXCHANGE_FN, EXCHANGE_WAIT_FN, vp]
    L.nb_enable_timing.argtypes = [vp, C.c_int]
    L.nb_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.nb_variant_name.argtypes = [vp]
    L.nb_variant_name.restype = C.c_char_p
    L.nb_diagnostics.argtypes = [vp, C.POINTER(C.c_double)]
    L.nb_multi_create.argtypes = [C.POINTER(nb_config), C.c_uint32, C.POINTER(C.c_int32), C.POINTER(vp)]
    L.nb_multi_destroy.argtypes = [vp]
    L.nb_multi_destroy.restype = None
    L.nb_multi_upload.argtypes = [vp, vp, vp, vp]
    L.nb_multi_set_params.argtypes = [vp, C.c_double, C.c_double]
    L.nb_multi_step.argtypes = [vp, C.c_uint32]
    L.nb_multi_download.argtypes = [vp, vp, vp, vp]
    L.nb_multi_sync.argtypes = [vp]
    L.nb_multi_last_error.argtypes = [vp]
    L.nb_multi_last_error.restype = C.c_char_p
    L.nb_multi_variant_name.argtypes = [vp]
    L.nb_multi_variant_name.restype = C.c_char_p
    L.nb_multi_diagnostics.argtypes = [vp, C.POINTER(C.c_double)]
    L.nb_multi_set_collective.argtypes = [vp, C.c_int]
    L.nb_multi_collective_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.nb_rccl_unique_id.argtypes = [vp]
    L.nb_rccl_attach.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32]
    L.nb_rccl_detach.argtypes = [vp]
    L.nb_rccl_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.nb_step_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_uint32)]
    L.nb_step_times2.argtypes = [vp, C.POINTER(nb_step_timing)]
    L.nb_integrate_pass.argtypes = [vp, C.c_uint32, C.POINTER(C.c_double)]
    L.nb_force_pass.argtypes = [vp, C.c_uint32, C.POINTER(C.c_double)]
    L.nb_shape_info.argtypes = [vp] + [C.POINTER(C.c_uint32)] * 4
    L.nb_plan_query.argtypes = [C.POINTER(nb_config), C.c_int, C.c_double, C.POINTER(nb_plan_info), C.POINTER(C.c_uint32), C.c_uint32]
    L.nb_frame_request.argtypes = [vp]
    L.nb_frame_acquire.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)),
                                   C.POINTER(C.c_uint64)]
    _lib = L
    return L


def abi_version():
    return load_library().nb_abi_version()


def abi_minor():
    return load_library().nb_abi_minor()


def device_count():
    return load_library().nb_device_count()


def rccl_unique_id():
    """ncclGetUniqueId through the engine: 128 bytes that rank 0 hands to every other rank."""
    L = load_library()
    buf = C.create_string_buffer(NB_RCCL_ID_BYTES)
    rc = L.nb_rccl_unique_id(buf)
    if rc != 0:
        raise NBodyError(rc, L.nb_last_error(None).decode())
    return buf.raw


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


SYMW_PLAN_WORDS = ("np", "nsb", "W", "total_hi", "total_lo", "n_hi", "H", "r_layer0", "t_layer0", "L", "zc")
SYM_RANK_PLAN_WORDS = ("np", "nsb", "total_hi", "total_lo", "n_hi", "H", "r_layer0", "rb_layer0", "t_layer0", "g0", "g1", "LA", "LB", "WA", "WB", "ups")
SYM_PLAN_WORDS = ("np", "nsb", "q", "total_hi", "total_lo", "n_hi", "H", "r_layer0", "t_layer0")


def plan_query(n, precision="f32", shard=None, force_variant=0, jsplit=0, flags=0, n_cu=256, clock_hz=2.4e9, device=-1,
               layer_budget_mib=0, sym_pass=0):
    """nb_plan_query: the launch plan nb_create would build -- the engine's planner run on the host alone (works without a
    GPU when n_cu and clock_hz are given; 0 means "as on the device").  Returns a dict: the shape digits, the j-partitions,
    and for the symmetric pass `plan` (the words the kernels receive, by name) and `tab` (first wave, resident layers per super-block)."""
    L = load_library()
    cfg = nb_config()
    cfg.struct_size = C.sizeof(nb_config)
    cfg.n = int(n)
    cfg.precision = NB_F64 if precision in ("f64", NB_F64, np.float64) else NB_F32
    cfg.device = device
    if shard is not None:
        cfg.shard_begin, cfg.shard_count = int(shard[0]), int(shard[1])
    cfg.force_variant, cfg.jsplit, cfg.flags = int(force_variant), int(jsplit), int(flags)
    cfg.layer_budget_mib = int(layer_budget_mib)
    info = nb_plan_info()
    info.struct_size = C.sizeof(nb_plan_info)
    info.sym_pass = int(sym_pass)
    rc = L.nb_plan_query(C.byref(cfg), int(n_cu), float(clock_hz), C.byref(info), None, 0)
    if rc != 0:
        raise NBodyError(rc, L.nb_last_error(None).decode())
    tab = np.zeros(info.tab_len, np.uint32)
    if info.tab_len:
        info.sym_pass = int(sym_pass)
        rc = L.nb_plan_query(C.byref(cfg), int(n_cu), float(clock_hz), C.byref(info), tab.ctypes.data_as(C.POINTER(C.c_uint32)), tab.size)
        if rc != 0:
            raise NBodyError(rc, L.nb_last_error(None).decode())
    out = {k: int(getattr(info, k)) for k, _ in nb_plan_info._fields_[1:16]}
    out["variant"] = info.variant.decode()
    out["flags"] = int(flags)
    out.update(passes=int(info.sym_passes), local=int(info.sym_local), pass_k_lo=int(info.sym_pass_k_lo), pass_k_hi=int(info.sym_pass_k_hi),
               pass_d0=int(info.sym_pass_d0))
    if info.sym:
        out["plan"] = dict(zip(SYMW_PLAN_WORDS if info.symw else SYM_PLAN_WORDS, (int(w) for w in info.sym_plan)))
        nsb = out["plan"]["nsb"]
        if info.sym_rank:
            # the rank form: two phases (own-row travelers first), {first A wave, A waves, first B wave, B waves} per super-block
            rp = dict(zip(SYM_RANK_PLAN_WORDS, (int(w) for w in info.sym_rank_plan)))
            ng = rp["g1"] - rp["g0"]
            out["rank_plan"] = rp
            out["rank_tab"] = tab[:4 * nsb].reshape(-1, 4)
            out["prefix_a"] = tab[4 * nsb:4 * nsb + ng + 1]
            out["prefix_b"] = tab[4 * nsb + ng + 1:4 * nsb + 2 * (ng + 1)]
            if rp["ups"] > 1:
                nch = rp["np"] // 64
                base = 4 * nsb + 2 * (ng + 1)
                out["spill_tab"] = tab[base:base + 2 * nch].reshape(-1, 2)
                out["spill_ids"] = tab[base + 2 * nch:]
            out["tab"] = out["rank_tab"][:, :2]
        else:
            if info.symw:
                # one {first wave, resident layers} pair per block of S rows: the nsb whole super-blocks of the ring, then the short block Z
                nsb = out["plan"]["np"] // (64 * int(info.ipl))
            out["tab"] = tab[:2 * nsb].reshape(-1, 2)
            if info.symw:
                # four words per physical wave: {first unit, end, resident layer of the super-block the range ends in, spill row}.  The ranges
                # partition the list; "order" lists the waves in list order ("positions"), "starts" the first unit of every position + the list's end
                W = out["plan"]["W"]
                out["waves"] = tab[2 * nsb:2 * nsb + 4 * W].reshape(-1, 4)
                out["order"] = np.argsort(out["waves"][:, 0], kind="stable")
                out["starts"] = np.concatenate([out["waves"][out["order"], 0], [out["plan"]["L"] * int(info.sym_ups)]]).astype(np.uint32)
                # (a wave's record ends where its OWN part ends: with whole sweeps and two waves per SIMD the last sweeps of an older
                # wave's range are left to a queue -- {first unit, resident layer | sweeps << 16} per queued piece, in queue order, behind the records)
                out["pieces"] = tab[2 * nsb + 4 * W:].reshape(-1, 2) if int(info.sym_ups) == 1 else np.zeros((0, 2), np.uint32)
        out["ups"], out["spill_rows"] = int(info.sym_ups), int(info.sym_spill_rows)
        if info.symw:
            out["plan"]["ups"] = int(info.sym_ups)
        if info.sym_ups > 1 and not info.sym_rank:
            # the spill rows (wave ranges cut inside sweeps): {first row, count} per traveler chunk, then the wave numbers in row order
            # (a wave's own row is word 3 of its record)
            ch = 128 if info.x == 1 else 64
            nch = out["plan"]["np"] // ch
            W = out["plan"]["W"]
            base = 2 * nsb + 4 * W
            out["spill_slot"] = out["waves"][:, 3]
            out["spill_tab"] = tab[base:base + 2 * nch].reshape(-1, 2)
            out["spill_ids"] = tab[base + 2 * nch:]
    return out


class Simulation:
    """One engine handle = the reference's (bodyBuffer, velBuffer, accelBuffer,
    uniforms, compute pipeline) bundle, nbody3d.js:13,179-204,296-311."""

    def __init__(self, n, precision="f32", eps2=None, device=-1, shard=None, stream=None, ext_bodies=None,
                 force_variant=0, jsplit=0, tile=0, flags=0, layer_budget_mib=0):
        L = load_library()
        self._L = L
        self.n = int(n)
        self.dtype = np.float64 if precision in ("f64", NB_F64, np.float64) else np.float32
        cfg = nb_config()
        cfg.struct_size = C.sizeof(nb_config)
        cfg.n = self.n
        cfg.precision = NB_F64 if self.dtype == np.float64 else NB_F32
        cfg.tile = tile
        cfg.eps2 = 0.0 if eps2 is None else float(eps2)
        cfg.device = device
        self.shard_begin, self.shard_count = (0, self.n) if shard is None else (int(shard[0]), int(shard[1]))
        if shard is not None:      # shard_count = 0 means "whole system, not a shard" (eligible for the fused step)
            cfg.shard_begin, cfg.shard_count = self.shard_begin, self.shard_count
        if stream is not None:
            cfg.ext_stream = stream if stream else None
            cfg.flags |= NB_FLAG_EXT_STREAM
        if ext_bodies is not None:
            cfg.ext_bodies = ext_bodies
        cfg.force_variant = force_variant
        cfg.jsplit = jsplit
        cfg.flags |= int(flags)
        cfg.layer_budget_mib = int(layer_budget_mib)
        h = C.c_void_p()
        rc = L.nb_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise NBodyError(rc, L.nb_last_error(None).decode())
        self._h = h
        self._hook = None
        self.dt = 0.0
        self.G = 0.0

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.nb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise NBodyError(rc, self._L.nb_last_error(self._h).decode())

    def _arr(self, a, name):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.size != 4 * self.n:
            raise ValueError("%s must hold 4*n = %d elements, got %d" % (name, 4 * self.n, a.size))
        return a

    # -- reference-shaped surface -------------------------------------------
    def init(self, bodies, vel, accel=None):
        """nbody3d.js:177-199: upload packed [x,y,z,m] / [vx,vy,vz,0]; accel zero."""
        b = self._arr(bodies, "bodies")
        v = self._arr(vel, "vel")
        a = None if accel is None else self._arr(accel, "accel")
        self._check(self._L.nb_upload(self._h, _ptr(b), _ptr(v), _ptr(a)))
        return self

    restore = init  # util.js:230-244 writes the same three arrays back

    def set_params(self, dt, G):
        self.dt, self.G = float(dt), float(G)
        self._check(self._L.nb_set_params(self._h, self.dt, self.G))

    def step(self, dt=None, G=None):
        """One frame's compute pass (nbody3d.js:470-490); dt <= 0 is a no-op (:474)."""
        if dt is not None or G is not None:
            self.set_params(self.dt if dt is None else dt, self.G if G is None else G)
        self._check(self._L.nb_step(self._h, 1))

    def simulate(self, nsteps, dt=None, G=None):
        if dt is not None or G is not None:
            self.set_params(self.dt if dt is None else dt, self.G if G is None else G)
        self._check(self._L.nb_step(self._h, int(nsteps)))

    def sync(self):
        self._check(self._L.nb_sync(self._h))

    def read(self, bodies=True, vel=True, accel=True):
        """util.js:163-178: fresh host copies of (bodies, vel, accel), shape (N,4).
        On a shard handle vel/accel rows outside the shard are zero."""
        out = [np.zeros((self.n, 4), self.dtype) if f else None for f in (bodies, vel, accel)]
        self._check(self._L.nb_download(self._h, _ptr(out[0]), _ptr(out[1]), _ptr(out[2])))
        return tuple(out)

    # -- multi-GPU / measurement / diagnostics ------------------------------
    def device_ptr(self, which):
        p = C.c_void_p()
        self._check(self._L.nb_device_ptr(self._h, {"bodies": 0, "vel": 1, "accel": 2}[which], C.byref(p)))
        return p.value

    def set_exchange(self, fn):
        """fn(bodies_dev_ptr, elem_size, n, shard_begin, shard_count, stream) -> 0."""
        if fn is None:
            self._hook = None
            self._check(self._L.nb_set_exchange(self._h, C.cast(None, EXCHANGE_FN), None))
            return

        def tramp(user, bodies, esz, n, sb, sc, stream):
            try:
                return int(fn(bodies, esz, n, sb, sc, stream) or 0)
            except Exception:  # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return -1

        self._hook = EXCHANGE_FN(tramp)  # keep alive
        self._check(self._L.nb_set_exchange(self._h, self._hook, None))

    def set_exchange_overlapped(self, begin, wait):
        """Two-phase hook (nb_set_exchange_overlapped): begin(bodies_ptr, esz, n, sb, sc,
        stream) starts the all-gather, wait(stream) makes the stream wait for it."""
        def t_begin(user, bodies, esz, n, sb, sc, stream):
            try:
                return int(begin(bodies, esz, n, sb, sc, stream) or 0)
            except Exception:
                import traceback
                traceback.print_exc()
                return -1

        def t_wait(user, stream):
            try:
                return int(wait(stream) or 0)
            except Exception:
                import traceback
                traceback.print_exc()
                return -1

        self._hook = (EXCHANGE_FN(t_begin), EXCHANGE_WAIT_FN(t_wait))  # keep alive
        self._check(self._L.nb_set_exchange_overlapped(self._h, self._hook[0], self._hook[1], None))

    def enable_timing(self, on=True):
        self._check(self._L.nb_enable_timing(self._h, 1 if on else 0))

    def kernel_times(self):
        """(avg force-kernel ms, avg integrate-kernel ms, launches) since last call."""
        f, g, c = C.c_double(), C.c_double(), C.c_uint32()
        self._check(self._L.nb_kernel_times(self._h, C.byref(f), C.byref(g), C.byref(c)))
        return f.value, g.value, c.value

    def step_times(self):
        """(force ms, integrate ms, native-RCCL exchange ms, launches) since last call."""
        f, g, x, c = C.c_double(), C.c_double(), C.c_double(), C.c_uint32()
        self._check(self._L.nb_step_times(self._h, C.byref(f), C.byref(g), C.byref(x), C.byref(c)))
        return f.value, g.value, x.value, c.value

    def step_breakdown(self):
        """The parts of a step as the engine stream runs them (nb_step_times2; averages in ms since the last call):
        dict(launches, force_ms, sym_reduce_ms, reduce_scatter_ms, integrate_ms, allgather_ms, span_ms, reduce_scatters,
        allgathers)."""
        t = nb_step_timing()
        t.struct_size = C.sizeof(nb_step_timing)
        self._check(self._L.nb_step_times2(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in nb_step_timing._fields_ if k != "struct_size"}

    def integrate_pass(self, reps):
        """Average ms of the integrate kernel alone over ``reps`` launches (measurement only:
        the particle state is garbage afterwards)."""
        ms = C.c_double()
        self._check(self._L.nb_integrate_pass(self._h, int(reps), C.byref(ms)))
        return ms.value

    def force_pass(self, reps):
        """Average ms of the force pass alone over ``reps`` runs (a rank-form shard: both phases + nb_sym_reduce); no communicator
        needed, the state is left untouched.  What one rank of an N-rank partition spends in its force pass, timed on one GPU."""
        ms = C.c_double()
        self._check(self._L.nb_force_pass(self._h, int(reps), C.byref(ms)))
        return ms.value

    # -- native RCCL collective (one process per GPU) ------------------------
    def rccl_attach(self, unique_id, nranks, rank, overlap=False):
        if len(unique_id) != NB_RCCL_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % NB_RCCL_ID_BYTES)
        buf = C.create_string_buffer(bytes(unique_id), NB_RCCL_ID_BYTES)
        self._check(self._L.nb_rccl_attach(self._h, buf, int(nranks), int(rank), NB_RCCL_OVERLAP if overlap else 0))

    def rccl_detach(self):
        self._check(self._L.nb_rccl_detach(self._h))

    def rccl_info(self):
        """(nranks, rank, rccl version code) of the attached communicator; zeros when none."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.nb_rccl_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- viewer frame feed (nbody3d.js:408-415,482-487; colour input :380) ----
    def request_frame(self):
        self._check(self._L.nb_frame_request(self._h))

    def frame(self, wait=True):
        """Newest finished frame: (bodies f32 (n,4), speed f32 (n,), step index) -- views of the
        engine's pinned host memory, valid until the fourth request_frame() after the one that
        produced them AND no longer than the handle itself (close() frees the memory: copy what
        must outlive it); None when wait=False and nothing has landed yet."""
        pb, ps, st = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.c_uint64()
        rc = self._L.nb_frame_acquire(self._h, 1 if wait else 0, C.byref(pb), C.byref(ps), C.byref(st))
        if rc == NB_NOT_READY:
            return None
        self._check(rc)
        b = np.ctypeslib.as_array(pb, shape=(self.n, 4))
        sp = np.ctypeslib.as_array(ps, shape=(self.n,))
        return b, sp, st.value

    @property
    def variant(self):
        return self._L.nb_variant_name(self._h).decode()

    def shape_info(self):
        """{jsplit, j_per_split, own_split0, own_splits}: the force pass's j-partitions and those that lie entirely
        inside this handle's own rows (what the overlapped exchange issues before waiting for the gather)."""
        v = [C.c_uint32() for _ in range(4)]
        self._check(self._L.nb_shape_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("jsplit", "j_per_split", "own_split0", "own_splits"), (x.value for x in v)))

    def diagnostics(self):
        """(kinetic, potential share, momentum[3]) of this handle's shard, fp64 on device."""
        out = (C.c_double * 5)()
        self._check(self._L.nb_diagnostics(self._h, out))
        return out[0], out[1], np.array(out[2:5])

    def energy_drift(self, steps, every):
        """Runs ``steps`` steps (parameters as set) and samples the total energy every ``every`` steps with the bookkeeping of
        SURVEY.md §8(c): the stored velocity lags the positions by one call (nbody3d.js:278-283), so KE(vel after call n) pairs
        with PE(positions BEFORE call n); E0 pairs the uploaded state.  Returns [|E_n - E0| / |E0|] at the sampled steps."""
        ke0, pe0, _ = self.diagnostics()
        e0 = ke0 + pe0
        out, done = [], 0
        while done < steps:
            k = min(int(every), steps - done)
            if k > 1:
                self.simulate(k - 1)
            _, pe_prev, _ = self.diagnostics()
            self.step()
            ke, _, _ = self.diagnostics()
            done += k
            out.append(abs((ke + pe_prev - e0) / e0))
        return out


class MultiSimulation:
    """Single-process multi-device handle (nb_multi_*): n_shards i-shards, one per
    entry of ``devices`` (default: round-robin over the visible GPUs; several
    shards may share a GPU), peer-copy all-gather after every step.  Same
    host-side surface as Simulation; arrays hold the unpadded n rows."""

    def __init__(self, n, n_shards, devices=None, precision="f32", eps2=None, force_variant=0, jsplit=0,
                 collective="peer"):
        L = load_library()
        self._L = L
        self.n, self.n_shards = int(n), int(n_shards)
        self.dtype = np.float64 if precision in ("f64", NB_F64, np.float64) else np.float32
        cfg = nb_config()
        cfg.struct_size = C.sizeof(nb_config)
        cfg.n = self.n
        cfg.precision = NB_F64 if self.dtype == np.float64 else NB_F32
        cfg.eps2 = 0.0 if eps2 is None else float(eps2)
        cfg.device = -1
        cfg.force_variant, cfg.jsplit = force_variant, jsplit
        dev = None
        if devices is not None:
            assert len(devices) == self.n_shards
            dev = (C.c_int32 * self.n_shards)(*devices)
        h = C.c_void_p()
        rc = L.nb_multi_create(C.byref(cfg), self.n_shards, dev, C.byref(h))
        if rc != 0:
            raise NBodyError(rc, L.nb_multi_last_error(None).decode())
        self._h = h
        self.dt = self.G = 0.0
        if collective != "peer":
            try:
                self.set_collective(collective)
            except Exception:
                self.close()
                raise

    def set_collective(self, mode):
        """'peer' (event-ordered device-to-device copies) or 'rccl' (ncclCommInitAll + grouped
        in-place ncclAllGather); bit-identical results."""
        self._check(self._L.nb_multi_set_collective(self._h, {"peer": NB_MULTI_PEER, "rccl": NB_MULTI_RCCL, "peer_overlap": NB_MULTI_PEER_OVERLAP}[mode]))

    def collective_info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.nb_multi_collective_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"mode": {NB_MULTI_RCCL: "rccl", NB_MULTI_PEER_OVERLAP: "peer_overlap"}.get(a.value, "peer"), "nranks": b.value, "rccl_version": c.value}

    def close(self):
        if getattr(self, "_h", None):
            self._L.nb_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise NBodyError(rc, self._L.nb_multi_last_error(self._h).decode())

    def _arr(self, a, name):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.size != 4 * self.n:
            raise ValueError("%s must hold 4*n = %d elements, got %d" % (name, 4 * self.n, a.size))
        return a

    def init(self, bodies, vel, accel=None):
        b, v = self._arr(bodies, "bodies"), self._arr(vel, "vel")
        a = None if accel is None else self._arr(accel, "accel")
        self._check(self._L.nb_multi_upload(self._h, _ptr(b), _ptr(v), _ptr(a)))
        return self

    restore = init

    def set_params(self, dt, G):
        self.dt, self.G = float(dt), float(G)
        self._check(self._L.nb_multi_set_params(self._h, self.dt, self.G))

    def step(self, dt=None, G=None):
        self.simulate(1, dt, G)

    def simulate(self, nsteps, dt=None, G=None):
        if dt is not None or G is not None:
            self.set_params(self.dt if dt is None else dt, self.G if G is None else G)
        self._check(self._L.nb_multi_step(self._h, int(nsteps)))

    def sync(self):
        self._check(self._L.nb_multi_sync(self._h))

    def read(self, bodies=True, vel=True, accel=True):
        out = [np.zeros((self.n, 4), self.dtype) if f else None for f in (bodies, vel, accel)]
        self._check(self._L.nb_multi_download(self._h, _ptr(out[0]), _ptr(out[1]), _ptr(out[2])))
        return tuple(out)

    def diagnostics(self):
        out = (C.c_double * 5)()
        self._check(self._L.nb_multi_diagnostics(self._h, out))
        return out[0], out[1], np.array(out[2:5])

    @property
    def variant(self):
        return self._L.nb_multi_variant_name(self._h).decode()
