"""ctypes binding of libgrappa_host.so (the C ABI of include/grappa_host.h): host-side tuple enumeration and graph features.
Built in-tree next to libgrappa_hip.so by `make -C grappa_amd/csrc` / `__graft_entry__.build()`."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgrappa_host.so")
ABI_VERSION = 3

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64 = C.POINTER(C.c_int64)
SIGNATURES = {
    "grappa_host_abi_version": (C.c_int, []),
    "grappa_topo_enumerate": (C.c_int, [C.c_int, _i32p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, _i64, _i64]),
    "grappa_degree_encoding": (C.c_int, [C.c_int, C.c_int, _i32p, _f32p]),
    "grappa_ring_encoding": (C.c_int, [C.c_int, C.c_int, _i32p, _f32p]),
    "grappa_components": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, _i32p]),
    "grappa_position_tables": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong]),
    "grappa_plan_build": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, _i32p, C.POINTER(C.c_void_p), _i32p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i32p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C grappa_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(LIB_PATH)
    # the version first: a stale library lacks the newer symbols, and its callers expect RuntimeError (they fall back to numpy), not AttributeError
    try:
        lib.grappa_host_abi_version.restype = C.c_int
        version = lib.grappa_host_abi_version()
    except AttributeError:
        version = None
    if version != ABI_VERSION:
        raise RuntimeError(f"libgrappa_host.so: ABI version {version}, this package needs {ABI_VERSION}: rebuild it (`make -C grappa_amd/csrc`)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"libgrappa_host.so does not export {name}: rebuild it (`make -C grappa_amd/csrc`)")
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _bonds32(bonds) -> np.ndarray:
    b = np.ascontiguousarray(np.asarray(bonds, dtype=np.int64).reshape(-1, 2))
    if b.size and (b.min() < 0 or b.max() >= 2 ** 31):
        raise ValueError("atom ids must be non-negative 32-bit integers")
    return np.ascontiguousarray(b.astype(np.int32))


def enumerate_tuples(bonds):
    """-> (angles (n,3) int32, propers (n,4) int32) in the reference's row order (include/grappa_host.h grappa_topo_enumerate)"""
    lib = load()
    b = _bonds32(bonds)
    na, npr = C.c_int64(0), C.c_int64(0)
    rc = lib.grappa_topo_enumerate(len(b), b, None, 0, None, 0, C.byref(na), C.byref(npr))
    if rc != 0:
        if len(b) and (b[:, 0] == b[:, 1]).any():
            raise AssertionError("Encountered self-bond")                  # the reference asserts (utils/tuple_indices.py:72)
        raise RuntimeError(f"grappa_topo_enumerate failed with status {rc}")
    angles, propers = np.empty((na.value, 3), dtype=np.int32), np.empty((npr.value, 4), dtype=np.int32)
    rc = lib.grappa_topo_enumerate(len(b), b, angles.ctypes.data_as(C.c_void_p), na.value, propers.ctypes.data_as(C.c_void_p), npr.value,
                                   C.byref(na), C.byref(npr))
    if rc != 0:
        raise RuntimeError(f"grappa_topo_enumerate failed with status {rc}")
    return angles, propers


def degree_encoding(n_atoms: int, bonds) -> np.ndarray:
    enc = np.empty((n_atoms, 6), dtype=np.float32)
    b = _bonds32(bonds)
    rc = load().grappa_degree_encoding(n_atoms, len(b), b, enc)
    if rc != 0:
        raise RuntimeError(f"grappa_degree_encoding failed with status {rc}")
    return enc


def ring_encoding(n_atoms: int, bonds) -> np.ndarray:
    enc = np.empty((n_atoms, 7), dtype=np.float32)
    b = _bonds32(bonds)
    rc = load().grappa_ring_encoding(n_atoms, len(b), b, enc)
    if rc != 0:
        raise RuntimeError(f"grappa_ring_encoding failed with status {rc}")
    return enc


def plan_build(N: int, src: np.ndarray, dst: np.ndarray, idx_levels):
    """index plan of a batched graph (include/grappa_host.h grappa_plan_build).  src / dst: int64 (E,); idx_levels: four int32 (T_l, arity)
    arrays (bond, angle, proper, improper).  -> dict of int32 arrays + max_degree; raises like grappa_amd.batch.BatchPlan"""
    lib = load()
    src = np.ascontiguousarray(src, dtype=np.int64)
    dst = np.ascontiguousarray(dst, dtype=np.int64)
    E = int(src.shape[0])
    arity = (2, 3, 4, 4)
    idx = [np.ascontiguousarray(a, dtype=np.int32).reshape(-1, s) for a, s in zip(idx_levels, arity)]
    T = np.array([a.shape[0] for a in idx], dtype=np.int32)
    out = {"indptr": np.empty(N + 1, np.int32), "indices": np.empty(E, np.int32), "rev": np.empty(E, np.int32), "inc_ptr": np.empty(N + 1, np.int32),
           "inc_code": np.empty(int(sum(s * t for s, t in zip(arity, T))), np.int32)}
    inv_ptr = [np.empty(N + 1, np.int32) for _ in range(4)]
    inv_rows = [np.empty(s * int(t), np.int32) for s, t in zip(arity, T)]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    arr = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data for x in xs])      # noqa: E731
    maxdeg, detail = C.c_int32(0), C.c_int32(0)
    rc = lib.grappa_plan_build(int(N), E, vp(src), vp(dst), T, arr(idx), out["indptr"], vp(out["indices"]), vp(out["rev"]), arr(inv_ptr), arr(inv_rows),
                               out["inc_ptr"], vp(out["inc_code"]), C.byref(maxdeg), C.byref(detail))
    if rc != 0:
        if detail.value == 1:
            raise ValueError("MolBatch: the n1 graph must contain both directions of every bond")
        if detail.value == 2:
            raise RuntimeError("There are 0-in-degree nodes in the graph (every atom must be bonded)")
        raise RuntimeError(f"grappa_plan_build failed with status {rc}")
    out["inv_ptr"], out["inv_rows"], out["max_degree"] = inv_ptr, inv_rows, int(maxdeg.value)
    return out


def components(n: int, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """label[a] = smallest atom index of a's connected component (include/grappa_host.h grappa_components)"""
    lib = load()
    src = np.ascontiguousarray(src, dtype=np.int64)
    dst = np.ascontiguousarray(dst, dtype=np.int64)
    label = np.empty(n, dtype=np.int32)
    rc = lib.grappa_components(int(n), int(src.shape[0]), src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), label)
    if rc != 0:
        raise RuntimeError(f"grappa_components failed with status {rc}")
    return label


def position_tables(N: int, idx: np.ndarray):
    """flat int32 array + (offset, size) of its six parts (include/grappa_host.h grappa_position_tables); idx: (T, s) int32"""
    lib = load()
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    T, s = idx.shape
    need = lib.grappa_position_tables(int(N), int(T), int(s), idx.ctypes.data_as(C.c_void_p), None, 0)
    if need < 0:
        raise RuntimeError(f"grappa_position_tables failed with status {need}")
    flat = np.empty(int(need), dtype=np.int32)
    rc = lib.grappa_position_tables(int(N), int(T), int(s), idx.ctypes.data_as(C.c_void_p), flat.ctypes.data_as(C.c_void_p), int(need))
    if rc < 0:
        raise RuntimeError(f"grappa_position_tables failed with status {rc}")
    sizes = [N * s, N + 1, N * s, T * s, s * N + 1, s * T]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 3) // 4 * 4
    return flat, list(zip(offs, sizes))
