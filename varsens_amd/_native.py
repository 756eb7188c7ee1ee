"""ctypes binding of libvarsens_hip.so (C ABI: include/varsens_hip.h).

This is the only door between the Python host code and the GPU.  There is no
CPU implementation behind it: if the shared library is missing, or no gfx950
device is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VARSENS_HIP_LIB") or os.path.join(_HERE, "libvarsens_hip.so")

ABI_VERSION = 8
COMM_ID_BYTES = 128

GEN_HALTON, GEN_SOBOL, GEN_RAW = 0, 1, 2
PERM_EMPTY = (1 << 64) - 1   # VS_PERM_EMPTY: vs_design_desc.perm_count of a rank without rows
DESIGN_SEED1_SHUFFLE = 1      # vs_design_desc.flags: the library supplies (and caches) the seed-1 shuffle itself
SCALE_IDENTITY, SCALE_LINEAR, SCALE_POWER = 0, 1, 2
OBJ_G_FUNCTION, OBJ_ISHIGAMI, OBJ_LINEAR, OBJ_POLYNOMIAL, OBJ_INVERT, OBJ_SOURCE = 1, 2, 3, 4, 5, 6
REDUCE_NAN_TRIM = 1
ORDER_FAST, ORDER_STRICT = 0, 1

KERNEL_NAMES = {
    1: "halton_fill", 2: "sobol_fill", 3: "sample_flat", 4: "eval", 5: "reduce_first",
    6: "fold", 7: "finish", 8: "nan_mask", 9: "second_order", 10: "second_fold", 11: "fused", 12: "allreduce",
    13: "values",
}

ACC_HEADER = 4


def acc_per_output(k):
    return 3 + 4 * k


def acc_slot(l):
    return 1 + 2 * l


def acc_len(k, l, max_ranks):
    return ACC_HEADER + l * acc_per_output(k) + max_ranks * acc_slot(l)


def gram_len(K):
    """doubles of the second-order sums of vs_reduce_second_order: SS = (C+D)(C+D)^T, TT = (C-D)(C-D)^T, K = k*l"""
    return 2 * int(K) * int(K)


def first_order_len(k, l):
    return (2 + 4 * k) * l


F_SHIFT = 11     # default log2 of the rows per row block of the f-value layout (VS_F_SHIFT, include/varsens_hip.h)


def to_device_layout(blocks, f_shift=F_SHIFT):
    """[2k+2][rows][l] (the reference's flat order, reshaped) -> the device's row-blocked f-values as one flat array
    (include/varsens_hip.h, "F-VALUE LAYOUT"): row blocks of 2^f_shift rows; inside a row block plane after plane, plane
    p = output * (2k+2) + block.  numpy on the host: the independent statement of the layout the tests hold the device's
    own conversion kernels (vs_upload_values / vs_download_values) and every reading kernel against."""
    b = np.asarray(blocks, dtype=np.float64)
    nb, rows, l = b.shape
    planes = np.ascontiguousarray(b.transpose(2, 0, 1)).reshape(l * nb, rows)
    step = 1 << f_shift
    if rows <= step:
        return planes.reshape(-1)
    return np.concatenate([planes[:, r0:r0 + step].reshape(-1) for r0 in range(0, rows, step)])


def from_device_layout(flat, k, l, rows, f_shift=F_SHIFT):
    """The inverse: the device's flat f-values -> [2k+2][rows][l]."""
    flat = np.asarray(flat).reshape(-1)
    nb = 2 * k + 2
    step = 1 << f_shift
    planes = np.empty((l * nb, rows), dtype=flat.dtype)
    at = 0
    for r0 in range(0, rows, step):
        w = min(step, rows - r0)
        planes[:, r0:r0 + w] = flat[at:at + l * nb * w].reshape(l * nb, w)
        at += l * nb * w
    return np.ascontiguousarray(planes.reshape(l, nb, rows).transpose(1, 2, 0))


class VarsensHipError(RuntimeError):
    """A call into libvarsens_hip.so failed."""


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("lds_bytes_per_cu", C.c_int32), ("clock_mhz", C.c_int32),
                ("hbm_bytes", C.c_uint64), ("f_shift", C.c_int32), ("reserved", C.c_int32)]


class ProfileRecord(C.Structure):
    _fields_ = [("kernel_id", C.c_int32), ("ms", C.c_float)]


class DesignDesc(C.Structure):
    _fields_ = [("generator", C.c_int32), ("k", C.c_int32), ("n", C.c_uint64), ("first_index", C.c_uint64),
                ("scale_mode", C.c_int32), ("flags", C.c_int32),
                ("lower", C.c_void_p), ("upper", C.c_void_p), ("perm", C.c_void_p), ("raw", C.c_void_p),
                ("sobol_directions", C.c_void_p), ("perm_begin", C.c_uint64), ("perm_count", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/varsens_hip.h declares
SIGNATURES = {
    "vs_abi_version": (C.c_int, []),
    "vs_last_error": (C.c_char_p, []),
    "vs_build_id": (C.c_char_p, []),
    "vs_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vs_context_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vs_context_destroy": (C.c_int, [C.c_void_p]),
    "vs_context_info": (C.c_int, [C.c_void_p, C.POINTER(DeviceInfo)]),
    "vs_sync": (C.c_int, [C.c_void_p]),
    "vs_context_set_eval_order": (C.c_int, [C.c_void_p, C.c_int]),
    "vs_stream_select": (C.c_int, [C.c_void_p, C.c_int]),
    "vs_event_record": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "vs_stream_wait_event": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "vs_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vs_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vs_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vs_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vs_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "vs_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vs_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "vs_profile_select": (C.c_int, [C.c_void_p, C.c_uint32]),
    "vs_profile_end": (C.c_int, [C.c_void_p, C.POINTER(ProfileRecord), C.c_int, C.POINTER(C.c_int)]),
    "vs_halton_fill": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]),
    "vs_sobol_fill": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "vs_seed1_permutation": (C.c_int, [C.c_void_p, C.c_uint64]),
    "vs_design_desc_validate": (C.c_int, [C.POINTER(DesignDesc), C.POINTER(C.c_uint64)]),
    "vs_design_create": (C.c_int, [C.c_void_p, C.POINTER(DesignDesc), C.POINTER(C.c_void_p)]),
    "vs_design_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vs_sample_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "vs_sample_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "vs_sample_rows_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]),
    "vs_sample_rows_wait": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vs_text_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p]),
    "vs_text_read": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                               C.POINTER(C.c_uint64)]),
    "vs_upload_values": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                   C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]),
    "vs_download_values": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]),
    "vs_objective_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_void_p)]),
    "vs_objective_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vs_objective_create_separable": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t,
                                                C.POINTER(C.c_void_p)]),
    "vs_objective_create_source": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_size_t,
                                             C.POINTER(C.c_void_p)]),
    "vs_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "vs_reduce_first_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    "vs_eval_reduce_supported": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "vs_eval_reduce_both_supported": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "vs_eval_reduce_first_second_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                                    C.c_int, C.c_void_p, C.c_void_p]),
    "vs_eval_reduce_first_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                             C.c_int, C.c_void_p]),
    "vs_finish_first_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    "vs_reduce_second_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p]),
    "vs_reduce_first_second_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vs_finish_second_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]),
    "vs_comm_unique_id": (C.c_int, [C.c_void_p]),
    "vs_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "vs_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "vs_comm_destroy": (C.c_int, [C.c_void_p]),
}

# dmabuf IPC is the only mode this pool's driver supports; RCCL between processes needs it, and the
# runtime reads the variable when it initialises (first HIP call), so it is set before the library loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_lib = None


def load_library(path=None):
    """dlopen the engine and bind every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.isfile(p) and path is None and not os.environ.get("VARSENS_HIP_LIB"):
        # not built yet (fresh checkout): build it now if a hipcc is around; there is nothing else to load
        try:
            from . import build as _build
            _build.build(verbose=False)
        except Exception as exc:
            raise VarsensHipError("libvarsens_hip.so is missing and could not be built: %s" % exc)
    if not os.path.isfile(p):
        raise VarsensHipError(
            "libvarsens_hip.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)" % p)
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.vs_abi_version() != ABI_VERSION:
        raise VarsensHipError("libvarsens_hip.so has ABI %d, binding expects %d" %
                              (lib.vs_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise VarsensHipError("%s (status %d)" % (load_library().vs_last_error().decode("utf-8", "replace"), rc))


def build_id():
    """vs_build_id() of the loaded library: what it was built from (varsens_amd.build.source_id)."""
    return load_library().vs_build_id().decode()


def text_write(path, array, delimiter=" "):
    """numpy.savetxt(path, array, delimiter=delimiter) -- the same bytes, formatted on several threads (vs_text_write;
    host code of the engine, no GPU needed).  1-D arrays are written one value per line, as numpy does."""
    a = np.ascontiguousarray(array, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    if a.ndim != 2:
        raise VarsensHipError("text_write takes a 1-D or 2-D array")
    _check(load_library().vs_text_write(os.fsencode(path), a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1],
                                        delimiter.encode("utf-8")))


def text_read(path, delimiter=None):
    """numpy.loadtxt(path, delimiter=delimiter, ndmin=2) -- the same values, parsed on several threads (vs_text_read)."""
    lib = load_library()
    rows, cols = C.c_uint64(0), C.c_uint64(0)
    d = None if not delimiter else delimiter.encode("utf-8")
    _check(lib.vs_text_read(os.fsencode(path), d, None, 0, C.byref(rows), C.byref(cols)))
    out = np.empty((rows.value, cols.value), dtype=np.float64)
    if out.size:
        _check(lib.vs_text_read(os.fsencode(path), d, out.ctypes.data_as(C.c_void_p), out.size, C.byref(rows),
                                C.byref(cols)))
    return out


def seed1_permutation(n):
    """The reference's seed-1 row shuffle as a permutation (host code of the engine, no GPU needed)."""
    out = np.empty(int(n), dtype=np.uint32)
    _check(load_library().vs_seed1_permutation(out.ctypes.data_as(C.c_void_p), int(n)))
    return out


def device_count():
    n = C.c_int(0)
    rc = load_library().vs_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _host(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


class DeviceBuffer(object):
    """A block of HBM owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(ctx.lib.vs_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def at(self, byte_offset):
        return C.c_void_p(self.ptr + int(byte_offset))

    def free(self):
        if self.ptr and self.ctx.handle:
            _check(self.ctx.lib.vs_free(self.ctx.handle, C.c_void_p(self.ptr)))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView(object):
    """A window [byte_offset, byte_offset + nbytes) of a DeviceBuffer; accepted wherever a buffer is.
    (The accumulator record and the second-order Gram sums live in ONE buffer so that a rank makes one
    all-reduce call; the kernels see two windows of it.)"""

    def __init__(self, buf, byte_offset, nbytes):
        if byte_offset < 0 or byte_offset + nbytes > buf.nbytes:
            raise VarsensHipError("view exceeds the device buffer")
        self.base, self.ctx = buf, buf.ctx
        self.ptr = buf.ptr + int(byte_offset)
        self.nbytes = int(nbytes)

    def at(self, byte_offset):
        return C.c_void_p(self.ptr + int(byte_offset))

    def free(self):
        pass    # the parent buffer owns the memory


class Context(object):
    """One GPU + one HIP stream (vs_context)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.handle = None
        h = C.c_void_p()
        _check(self.lib.vs_context_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._handles = []      # weak references to the designs / objectives made through this context

    # -- plumbing
    def close(self):
        """Destroys the context; designs and objectives made through it are released first (a design belongs to the
        context that made it, and vs_context_destroy refuses while any is alive)."""
        if self.handle:
            for ref in list(self._handles):
                h = ref()
                if h is not None:
                    h.free()
            self._handles = []
            self.lib.vs_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        d = DeviceInfo()
        _check(self.lib.vs_context_info(self.handle, C.byref(d)))
        return dict(name=d.name.decode(), arch=d.arch.decode(), compute_units=d.compute_units,
                    wavefront_size=d.wavefront_size, lds_bytes_per_cu=d.lds_bytes_per_cu,
                    clock_mhz=d.clock_mhz, hbm_bytes=d.hbm_bytes, f_shift=d.f_shift)

    def sync(self):
        _check(self.lib.vs_sync(self.handle))

    def set_eval_order(self, order):
        """'fast' (default): scan kernels, reassociated k-term folds; 'strict': left-to-right folds, bit-identical
        to numpy.prod over the factors (include/varsens_hip.h, vs_context_set_eval_order)."""
        if order not in ("fast", "strict"):
            raise VarsensHipError("order must be 'fast' or 'strict'")
        _check(self.lib.vs_context_set_eval_order(self.handle, ORDER_STRICT if order == "strict" else ORDER_FAST))
        self.eval_order = order

    def eval_order_scope(self, order):
        """Context manager: evaluate in `order` inside, the context's own order again outside (an Objective whose
        values were deferred evaluates in the order its context had when it was made)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            before = getattr(self, "eval_order", "fast")
            if before != order:
                self.set_eval_order(order)
            try:
                yield
            finally:
                if before != order:
                    self.set_eval_order(before)
        return scope()

    def stream_select(self, which):
        _check(self.lib.vs_stream_select(self.handle, int(which)))

    def event_record(self, slot, stream):
        _check(self.lib.vs_event_record(self.handle, int(slot), int(stream)))

    def stream_wait_event(self, stream, slot):
        _check(self.lib.vs_stream_wait_event(self.handle, int(stream), int(slot)))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, array, dtype=np.float64):
        a, p = _host(array, dtype)
        buf = DeviceBuffer(self, a.nbytes)
        _check(self.lib.vs_upload(self.handle, C.c_void_p(buf.ptr), p, a.nbytes))
        return buf

    def upload_into(self, buf, array, dtype=np.float64, byte_offset=0):
        a, p = _host(array, dtype)
        if byte_offset + a.nbytes > buf.nbytes:
            raise VarsensHipError("upload exceeds the device buffer")
        _check(self.lib.vs_upload(self.handle, buf.at(byte_offset), p, a.nbytes))

    def download(self, buf, shape, dtype=np.float64, byte_offset=0):
        out = np.empty(shape, dtype=dtype)
        if byte_offset + out.nbytes > buf.nbytes:
            raise VarsensHipError("download exceeds the device buffer")
        _check(self.lib.vs_download(self.handle, out.ctypes.data_as(C.c_void_p), buf.at(byte_offset), out.nbytes))
        return out

    @property
    def f_shift(self):
        """log2 of the rows per row block of this context's f-value layout"""
        if getattr(self, "_f_shift", None) is None:
            self._f_shift = int(self.info()["f_shift"])
        return self._f_shift

    def upload_blocks(self, blocks):
        """[2k+2][rows][l] host values -> a device buffer in the device layout, laid out by numpy on the host (test and
        tool helper: the product path is upload_values, which converts on the device)."""
        return self.upload(to_device_layout(blocks, self.f_shift))

    def download_blocks(self, f_buf, k, l, rows):
        """The device buffer's f-values as [2k+2][rows][l], un-laid by numpy on the host (test and tool helper)."""
        return from_device_layout(self.download(f_buf, (2 * (k + 1) * rows * l,)), k, l, rows, self.f_shift)

    def upload_values(self, values, n, k, row_begin, rows, f_buf, divide=None):
        """Objective.load on the device (vs_upload_values): `values` is the reference's flat (2n(k+1), l) array; rows
        [row_begin, row_begin + rows) of every block land in `f_buf` in the planar layout, divided by `divide` (None, a
        scalar or l divisors).  Returns the number of shard rows Objective.load would delete (NaN in output 0)."""
        v = np.asarray(values)
        if v.dtype != np.float64 or not v.flags.c_contiguous:
            v = np.ascontiguousarray(v, dtype=np.float64)
        if v.ndim == 1:
            v = v.reshape(-1, 1)
        l = v.shape[1]
        if v.shape[0] != 2 * n * (k + 1):
            raise VarsensHipError("values must have 2n(k+1) = %d rows, got %d" % (2 * n * (k + 1), v.shape[0]))
        if 8 * 2 * (k + 1) * rows * l > f_buf.nbytes:
            raise VarsensHipError("values exceed the device buffer")
        dptr, nd = None, 0
        if divide is not None:
            d = np.ascontiguousarray(np.asarray(divide, dtype=np.float64).ravel())
            if d.size not in (1, l):
                raise VarsensHipError("divide must be a scalar or hold one divisor per output")
            dptr, nd = d.ctypes.data_as(C.c_void_p), int(d.size)
        lost = C.c_uint64(0)
        _check(self.lib.vs_upload_values(self.handle, v.ctypes.data_as(C.c_void_p), int(n), int(k), int(l),
                                         int(row_begin), int(rows), dptr, nd, C.c_void_p(f_buf.ptr), C.byref(lost)))
        return int(lost.value)

    def download_values(self, f_buf, rows, k, l):
        """Objective.flat() of the shard (vs_download_values): (2(k+1) rows, l) in the reference's order."""
        if 8 * 2 * (k + 1) * rows * l > f_buf.nbytes:
            raise VarsensHipError("download exceeds the device buffer")
        out = np.empty((2 * (k + 1) * int(rows), int(l)), dtype=np.float64)
        _check(self.lib.vs_download_values(self.handle, C.c_void_p(f_buf.ptr), int(rows), int(k), int(l),
                                           out.ctypes.data_as(C.c_void_p)))
        return out

    def sample_rows(self, design, row_begin, row_count, k):
        """Rows of Sample.flat() straight into a host array (vs_sample_rows)."""
        out = np.empty((int(row_count), int(k)), dtype=np.float64)
        _check(self.lib.vs_sample_rows(self.handle, design.h, int(row_begin), int(row_count),
                                       out.ctypes.data_as(C.c_void_p)))
        return out

    def sample_rows_begin(self, design, row_begin, row_count, slot):
        _check(self.lib.vs_sample_rows_begin(self.handle, design.h, int(row_begin), int(row_count), int(slot)))

    def sample_rows_wait(self, slot, row_count, k):
        """The rows begun on `slot` as a READ-ONLY view of the library's pinned memory -- valid until the next
        sample_rows_begin on the same slot."""
        p = C.c_void_p()
        _check(self.lib.vs_sample_rows_wait(self.handle, int(slot), C.byref(p)))
        buf = (C.c_double * (int(row_count) * int(k))).from_address(p.value)
        view = np.frombuffer(buf, dtype=np.float64).reshape(int(row_count), int(k))
        view.setflags(write=False)
        return view

    def memset(self, buf, byte=0):
        _check(self.lib.vs_memset(self.handle, C.c_void_p(buf.ptr), int(byte), buf.nbytes))

    def accumulate(self, dst_buf, src_buf, count):
        """dst[i] += src[i] for `count` doubles, on the device."""
        if 8 * int(count) > min(dst_buf.nbytes, src_buf.nbytes):
            raise VarsensHipError("accumulate exceeds a device buffer")
        _check(self.lib.vs_accumulate(self.handle, C.c_void_p(dst_buf.ptr), C.c_void_p(src_buf.ptr), int(count)))

    def profile_begin(self, capacity=4096, only=None):
        """Bracket the launches that follow with HIP events; `only`: names (KERNEL_NAMES) of the kernels to bracket,
        default all.  Every bracketed launch costs two event packets on its stream (vs_profile_select)."""
        mask = 0
        if only is not None:
            ids = {name: kid for kid, name in KERNEL_NAMES.items()}
            for name in only:
                if name not in ids:
                    raise VarsensHipError("unknown kernel name %r (known: %s)" % (name, sorted(ids)))
                mask |= 1 << ids[name]
        _check(self.lib.vs_profile_select(self.handle, mask))
        self._prof_capacity = int(capacity)
        _check(self.lib.vs_profile_begin(self.handle, self._prof_capacity))

    def profile_end(self):
        recs = (ProfileRecord * self._prof_capacity)()
        n = C.c_int(0)
        _check(self.lib.vs_profile_end(self.handle, recs, self._prof_capacity, C.byref(n)))
        return [(KERNEL_NAMES.get(recs[i].kernel_id, str(recs[i].kernel_id)), float(recs[i].ms))
                for i in range(n.value)]

    # -- generators
    def halton_fill(self, k, start, count):
        buf = DeviceBuffer(self, 8 * int(count) * int(k))
        _check(self.lib.vs_halton_fill(self.handle, int(k), int(start), int(count), C.c_void_p(buf.ptr)))
        return buf

    def sobol_fill(self, k, directions, start, count):
        d, dp = _host(directions, np.uint32)
        if d.shape != (k, 32):
            raise VarsensHipError("direction table must have shape (k, 32)")
        buf = DeviceBuffer(self, 8 * int(count) * int(k))
        _check(self.lib.vs_sobol_fill(self.handle, int(k), dp, int(start), int(count), C.c_void_p(buf.ptr)))
        return buf

    # -- design / objective handles
    def _track(self, handle):
        import weakref
        if len(self._handles) > 256:    # drop the references whose handles are gone
            self._handles = [r for r in self._handles if r() is not None and r().h]
        self._handles.append(weakref.ref(handle))
        return handle

    def design_create(self, generator, k, n, first_index=0, scale_mode=SCALE_IDENTITY, lower=None, upper=None,
                      perm=None, raw=None, sobol_directions=None, perm_begin=0, seed1_shuffle=None):
        """``perm``: the whole seed-1 permutation (n entries), or -- with ``perm_begin`` -- the entries of a
        rank's own rows [perm_begin, perm_begin + len(perm)) only.  ``seed1_shuffle=(begin, count)`` instead of
        ``perm``: the library supplies the seed-1 shuffle of those rows itself and keeps its device copy in the
        context (VS_DESIGN_SEED1_SHUFFLE)."""
        keep = []

        def ptr(a, dtype, shape=None):
            if a is None:
                return None
            arr, p = _host(a, dtype)
            if shape is not None and arr.shape != shape:
                raise VarsensHipError("array of shape %s expected, got %s" % (shape, arr.shape))
            keep.append(arr)
            return p
        desc = DesignDesc()
        desc.generator = int(generator)
        desc.k = int(k)
        desc.n = int(n)
        desc.first_index = int(first_index)
        desc.scale_mode = int(scale_mode)
        desc.lower = ptr(lower, np.float64, (k,))
        desc.upper = ptr(upper, np.float64, (k,))
        desc.perm = ptr(perm, np.uint32)
        if perm is not None:
            count = keep[-1].size
            if keep[-1].ndim != 1 or perm_begin < 0 or perm_begin + count > n:
                raise VarsensHipError("perm must hold the shuffle entries of rows [perm_begin, perm_begin+len) of n")
            if count == 0:      # an empty shard: 0 would mean "all n rows" to the C ABI
                desc.perm, count = ptr(np.zeros(1, np.uint32), np.uint32), PERM_EMPTY
            desc.perm_begin, desc.perm_count = int(perm_begin), int(count)
        if seed1_shuffle is not None:
            if perm is not None:
                raise VarsensHipError("seed1_shuffle and perm exclude each other")
            begin, count = int(seed1_shuffle[0]), int(seed1_shuffle[1])
            if begin < 0 or count < 0 or begin + count > n:
                raise VarsensHipError("seed1_shuffle=(begin, count) must lie inside the n rows")
            desc.flags = DESIGN_SEED1_SHUFFLE
            desc.perm_begin, desc.perm_count = begin, (count if count else PERM_EMPTY)
        desc.raw = ptr(raw, np.float64, (2 * n, k))
        desc.sobol_directions = ptr(sobol_directions, np.uint32, (k, 32))
        h = C.c_void_p()
        _check(self.lib.vs_design_create(self.handle, C.byref(desc), C.byref(h)))
        return self._track(Handle(self, h, self.lib.vs_design_destroy))

    def objective_create(self, kind, k, l, params=None):
        if params is None or len(params) == 0:
            p, n = None, 0
        else:
            arr, p = _host(np.asarray(params, dtype=np.float64).ravel(), np.float64)
            n = arr.size
        h = C.c_void_p()
        _check(self.lib.vs_objective_create(self.handle, int(kind), int(k), int(l), p, n, C.byref(h)))
        return self._track(Handle(self, h, self.lib.vs_objective_destroy))

    def objective_create_source(self, k, l, body, params=None):
        if params is None or len(params) == 0:
            p, n = None, 0
        else:
            arr, p = _host(np.asarray(params, dtype=np.float64).ravel(), np.float64)
            n = arr.size
        h = C.c_void_p()
        _check(self.lib.vs_objective_create_source(self.handle, int(k), int(l), body.encode("utf-8"), p, n,
                                                   C.byref(h)))
        return self._track(Handle(self, h, self.lib.vs_objective_destroy))

    def objective_create_separable(self, k, term_body, fold_sum, params=None):
        if params is None or len(params) == 0:
            p, n = None, 0
        else:
            arr, p = _host(np.asarray(params, dtype=np.float64).ravel(), np.float64)
            n = arr.size
        h = C.c_void_p()
        _check(self.lib.vs_objective_create_separable(self.handle, int(k), term_body.encode("utf-8"),
                                                      1 if fold_sum else 0, p, n, C.byref(h)))
        return self._track(Handle(self, h, self.lib.vs_objective_destroy))

    # -- compute
    def sample_flat(self, design, row_begin, row_count, out_buf):
        _check(self.lib.vs_sample_flat(self.handle, design.h, int(row_begin), int(row_count),
                                       C.c_void_p(out_buf.ptr)))

    def eval(self, design, objective, row_begin, row_count, f_buf):
        _check(self.lib.vs_eval(self.handle, design.h, objective.h, int(row_begin), int(row_count),
                                C.c_void_p(f_buf.ptr)))

    def reduce_first_order(self, f_buf, rows, k, l, acc_buf, flags=0, rank=0, max_ranks=1):
        _check(self.lib.vs_reduce_first_order(self.handle, C.c_void_p(f_buf.ptr), int(rows), int(k), int(l),
                                              int(flags), int(rank), int(max_ranks), C.c_void_p(acc_buf.ptr)))

    def eval_reduce_supported(self, design, objective):
        """True when the fused evaluate+reduce kernels cover this (design, objective)."""
        flag = C.c_int(0)
        _check(self.lib.vs_eval_reduce_supported(self.handle, design.h, objective.h, C.byref(flag)))
        return bool(flag.value)

    def eval_reduce_first_order(self, design, objective, row_begin, row_count, acc_buf, rank=0, max_ranks=1):
        _check(self.lib.vs_eval_reduce_first_order(self.handle, design.h, objective.h, int(row_begin), int(row_count),
                                                   int(rank), int(max_ranks), C.c_void_p(acc_buf.ptr)))

    def eval_reduce_both_supported(self, design, objective):
        """True when the fused evaluate + first/second-order kernels cover this (design, objective)."""
        flag = C.c_int(0)
        _check(self.lib.vs_eval_reduce_both_supported(self.handle, design.h, objective.h, C.byref(flag)))
        return bool(flag.value)

    def eval_reduce_first_second_order(self, design, objective, row_begin, row_count, acc_buf, gram_buf, rank=0,
                                       max_ranks=1):
        _check(self.lib.vs_eval_reduce_first_second_order(self.handle, design.h, objective.h, int(row_begin),
                                                          int(row_count), int(rank), int(max_ranks),
                                                          C.c_void_p(acc_buf.ptr), C.c_void_p(gram_buf.ptr)))

    def finish_first_order(self, acc_buf, n, k, l, out_buf, max_ranks=1):
        _check(self.lib.vs_finish_first_order(self.handle, C.c_void_p(acc_buf.ptr), int(n), int(k), int(l),
                                              int(max_ranks), C.c_void_p(out_buf.ptr)))

    def reduce_second_order(self, f_buf, rows, k, l, gram_buf, flags=0):
        _check(self.lib.vs_reduce_second_order(self.handle, C.c_void_p(f_buf.ptr), int(rows), int(k), int(l),
                                               int(flags), C.c_void_p(gram_buf.ptr)))

    def reduce_first_second_order(self, f_buf, rows, k, l, acc_buf, gram_buf, flags=0, rank=0, max_ranks=1):
        """Both reductions of compute_varsens in one pass over the f-values where the shape allows."""
        _check(self.lib.vs_reduce_first_second_order(self.handle, C.c_void_p(f_buf.ptr), int(rows), int(k), int(l),
                                                     int(flags), int(rank), int(max_ranks), C.c_void_p(acc_buf.ptr),
                                                     C.c_void_p(gram_buf.ptr)))

    def finish_second_order(self, gram_buf, first_buf, n, k, l, s2_buf, s2n_buf):
        _check(self.lib.vs_finish_second_order(self.handle, C.c_void_p(gram_buf.ptr), C.c_void_p(first_buf.ptr),
                                               int(n), int(k), int(l), C.c_void_p(s2_buf.ptr),
                                               C.c_void_p(s2n_buf.ptr)))

    # -- multi-GPU
    def comm_unique_id(self):
        raw = C.create_string_buffer(COMM_ID_BYTES)
        _check(self.lib.vs_comm_unique_id(raw))
        return raw.raw

    def comm_init(self, unique_id, nranks, rank):
        if len(unique_id) != COMM_ID_BYTES:
            raise VarsensHipError("unique id must be %d bytes" % COMM_ID_BYTES)
        _check(self.lib.vs_comm_init(self.handle, C.c_char_p(unique_id), int(nranks), int(rank)))

    def comm_allreduce(self, buf, count):
        _check(self.lib.vs_comm_allreduce(self.handle, C.c_void_p(buf.ptr), int(count)))

    def comm_destroy(self):
        _check(self.lib.vs_comm_destroy(self.handle))


class Handle(object):
    """A vs_design / vs_objective owned by a Context."""

    def __init__(self, ctx, h, destroy):
        self.ctx, self.h, self._destroy = ctx, h, destroy

    def free(self):
        if self.h and self.ctx.handle:
            self._destroy(self.ctx.handle, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
