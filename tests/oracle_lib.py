"""ctypes wrapper around oracle/liboracle.so -- the CPU checker (test infrastructure only)."""
import ctypes as C

import numpy as np


class OrcInput(C.Structure):
    _fields_ = [("n_sites", C.c_size_t), ("n_ids", C.c_uint), ("alleles", C.c_void_p),
                ("n_ref", C.c_void_p), ("n_alt", C.c_void_p), ("f_override", C.c_void_p),
                ("eps", C.c_double), ("max_cov", C.c_uint), ("window", C.c_uint)]


class Oracle:
    def __init__(self, path):
        self.lib = lib = C.CDLL(path)
        lib.orc_nck_table.restype = C.c_void_p
        lib.orc_nck_table.argtypes = [C.c_uint]
        lib.orc_pDgG.restype = C.c_double
        lib.orc_pDgG.argtypes = [C.c_void_p, C.c_uint, C.c_double, C.c_uint, C.c_uint, C.c_uint]
        lib.orc_pDgf.restype = C.c_double
        lib.orc_pDgf.argtypes = [C.c_double] * 4
        lib.orc_pDgIBD1.restype = C.c_double
        lib.orc_pDgIBD1.argtypes = [C.c_uint, C.c_uint] + [C.c_double] * 4
        lib.orc_compare.restype = C.c_size_t
        lib.orc_compare.argtypes = [C.POINTER(OrcInput), C.c_uint, C.c_void_p, C.c_size_t, C.c_int,
                                    C.c_int] + [C.c_void_p] * 6
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]

    def pdg(self, eps, max_cov, r, a):
        t = self.lib.orc_nck_table(max_cov)
        try:
            return [self.lib.orc_pDgG(t, max_cov, eps, g, r, a) for g in (0, 1, 2)]
        finally:
            self._libc.free(t)

    def nck(self, n):
        t = self.lib.orc_nck_table(n)
        arr = np.ctypeslib.as_array(C.cast(t, C.POINTER(C.c_ulong)), shape=((n + 1) * (n + 1),)).copy()
        self._libc.free(t)
        return arr.reshape(n + 1, n + 1)

    def compare(self, alleles, n_ref, n_alt, target, *, window=100, eps=0.02, max_cov=20,
                refids=None, pu_id=-1, ld=True, f_override=None):
        """alleles: uint8 [L][2N]; returns dict of per-site and per-window arrays."""
        alleles = np.ascontiguousarray(alleles, dtype=np.uint8)
        L, two_n = alleles.shape
        n_ref = np.ascontiguousarray(n_ref, dtype=np.uint8)
        n_alt = np.ascontiguousarray(n_alt, dtype=np.uint8)
        fo = None if f_override is None else np.ascontiguousarray(f_override, dtype=np.float64)
        inp = OrcInput(L, two_n // 2, alleles.ctypes.data, n_ref.ctypes.data, n_alt.ctypes.data,
                       None if fo is None else fo.ctypes.data, eps, max_cov, window)
        rid = None if refids is None else np.ascontiguousarray(refids, dtype=np.int32)
        max_win = L // window + 2
        af = np.empty(L)
        site = np.empty((L, 3))
        win = np.empty((max_win, 3))
        first = np.empty(max_win, dtype=np.uint32)
        last = np.empty(max_win, dtype=np.uint32)
        ns = np.empty(max_win, dtype=np.uint32)
        n = self.lib.orc_compare(C.byref(inp), int(target), None if rid is None else rid.ctypes.data,
                                 0 if rid is None else len(rid), int(pu_id), int(bool(ld)),
                                 af.ctypes.data, site.ctypes.data, win.ctypes.data,
                                 first.ctypes.data, last.ctypes.data, ns.ctypes.data)
        return dict(af=af, site=site, win=win[:n].copy(), first=first[:n].copy(),
                    last=last[:n].copy(), nsites=ns[:n].copy())
