"""ctypes binding of include/ibdgem_hip.h (one class per ibdg_ctx).

Mirrors the C ABI one-to-one; no arithmetic happens here.  If
libibdgem_hip.so is missing or a call fails, an EngineError is raised -- there
is no fallback path.
"""
import ctypes as C
import os

import numpy as np

# IBDG_LIB: another build of the same ABI (used to A/B kernel variants on one box)
LIB_PATH = os.environ.get("IBDG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libibdgem_hip.so")

# every symbol include/ibdgem_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "ibdg_abi_version": (C.c_int, []),
    "ibdg_device_count": (C.c_int, []),
    "ibdg_create": (_P, [C.c_int, C.c_double, C.c_uint]),
    "ibdg_destroy": (None, [_P]),
    "ibdg_last_error": (C.c_char_p, [_P]),
    "ibdg_pdg_table": (C.c_int, [C.c_double, C.c_uint, _P]),
    "ibdg_pdg_ibd0": (C.c_double, [C.c_double] * 4),
    "ibdg_pdg_ibd1": (C.c_double, [C.c_uint, C.c_uint] + [C.c_double] * 4),
    "ibdg_row_words": (C.c_size_t, [C.c_uint]),
    "ibdg_pack_alleles": (None, [_P, C.c_uint, _P]),
    "ibdg_pack_hap_text": (C.c_int, [C.c_char_p, C.c_uint, _P]),
    "ibdg_upload_panel": (C.c_int, [_P, _P, C.c_size_t, C.c_uint]),
    "ibdg_upload_panel_dev": (C.c_int, [_P, _P, C.c_size_t, C.c_uint]),
    "ibdg_upload_sites": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint]),
    "ibdg_upload_sites_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint]),
    "ibdg_upload_ms": (C.c_int, [_P, _P]),
    "ibdg_host_alloc": (_P, [C.c_size_t]),
    "ibdg_host_free": (None, [_P]),
    "ibdg_num_sites": (C.c_size_t, [_P]),
    "ibdg_num_windows": (C.c_size_t, [_P]),
    "ibdg_num_targets": (C.c_size_t, [_P]),
    "ibdg_upload_panel_fd": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_size_t, C.c_uint]),
    "ibdg_get_windows": (C.c_int, [_P, _P, _P, _P]),
    "ibdg_run": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int, C.c_int]),
    "ibdg_get_site_af": (C.c_int, [_P, _P]),
    "ibdg_get_site_ll": (C.c_int, [_P, C.c_size_t, _P]),
    "ibdg_get_window_ll": (C.c_int, [_P, C.c_size_t, _P]),
    "ibdg_get_window_ll_all": (C.c_int, [_P, _P]),
    "ibdg_get_alt_counts": (C.c_int, [_P, C.c_size_t, C.c_size_t, _P]),
    "ibdg_last_run_ms": (C.c_int, [_P, _P]),
    "ibdg_run_ms": (C.c_int, [_P, C.c_uint, _P]),
    "ibdg_run_kernel_ms": (C.c_int, [_P, C.c_uint, _P]),
    "ibdg_last_ld_variant": (C.c_int, [_P]),
    "ibdg_ld_layout": (C.c_int, [_P]),
    "ibdg_last_count_unit": (C.c_int, [_P]),
    "ibdg_set_option": (C.c_int, [_P, C.c_char_p, C.c_long]),
    "ibdg_set_background_order": (C.c_int, [_P, _P, C.c_size_t]),
    "ibdg_selftest": (C.c_int, [C.c_char_p]),
    "ibdg_sync": (C.c_int, [_P]),
}


class EngineError(RuntimeError):
    pass


_lib = None


def load_library(path=LIB_PATH):
    """dlopen the engine and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    if not os.path.exists(path):
        raise EngineError(f"{path} not found: build it with `make -C ibdgem_amd/csrc` "
                          "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        if path != LIB_PATH and not hasattr(lib, name):
            continue                   # an older build of the ABI loaded for an A/B comparison
        fn = getattr(lib, name)        # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path == LIB_PATH:
        _lib = lib
    return lib


class PinnedArray:
    """A numpy array over page-locked memory from ibdg_host_alloc (freed by close() or the collector)."""

    def __init__(self, shape, dtype):
        lib = load_library()
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * self.dtype.itemsize
        self.ptr = lib.ibdg_host_alloc(n)
        if not self.ptr:
            raise EngineError("ibdg_host_alloc failed")
        self._lib = lib
        buf = (C.c_char * max(n, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._lib.ibdg_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_alleles(alleles):
    """uint8 [L][2N] (0/1) -> packed rows uint64 [L][row_words] via ibdg_pack_alleles."""
    lib = load_library()
    alleles = np.ascontiguousarray(alleles, dtype=np.uint8)
    L, two_n = alleles.shape
    n_ids = two_n // 2
    rw = lib.ibdg_row_words(n_ids)
    out = np.zeros((L, rw), dtype=np.uint64)
    for i in range(L):
        lib.ibdg_pack_alleles(alleles[i].ctypes.data, n_ids, out[i].ctypes.data)
    return out


def pack_alleles_fast(alleles):
    """Same layout as pack_alleles, vectorised in numpy (test/bench data preparation only)."""
    alleles = np.ascontiguousarray(alleles, dtype=np.uint8)
    L, two_n = alleles.shape
    n_ids = two_n // 2
    chunks = (n_ids + 63) // 64
    pad = np.zeros((L, chunks * 64, 2), dtype=np.uint8)
    pad[:, :n_ids, :] = alleles.reshape(L, n_ids, 2)
    # [L][chunk][bit][plane] -> [L][chunk][plane][bit]
    bits = pad.reshape(L, chunks, 64, 2).transpose(0, 1, 3, 2)
    by = np.packbits(bits, axis=-1, bitorder="little")          # [L][chunk][plane][8 bytes]
    return np.ascontiguousarray(by).view("<u8").reshape(L, chunks * 2)


class Engine:
    def __init__(self, device=0, epsilon=0.02, max_cov=20, lib_path=None):
        self.lib = load_library(lib_path) if lib_path else load_library()
        self.ctx = self.lib.ibdg_create(device, epsilon, max_cov)
        if not self.ctx:
            raise EngineError(self.lib.ibdg_last_error(None).decode())
        self.n_ids = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.ibdg_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(self.lib.ibdg_last_error(self.ctx).decode() or f"error {rc}")

    def set_option(self, name, value):
        self._chk(self.lib.ibdg_set_option(self.ctx, name.encode(), int(value)))

    def upload_panel(self, rows, n_ids):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        assert rows.ndim == 2 and rows.shape[1] == self.lib.ibdg_row_words(n_ids)
        self._chk(self.lib.ibdg_upload_panel(self.ctx, rows.ctypes.data, rows.shape[0], n_ids))
        self.n_ids = n_ids

    def upload_panel_fd(self, fd, offset, n_rows, n_ids):
        """Packed rows from an open file (n_rows x ibdg_row_words(n_ids) x 8 bytes from byte `offset`)."""
        self._chk(self.lib.ibdg_upload_panel_fd(self.ctx, int(fd), int(offset), int(n_rows), int(n_ids)))
        self.n_ids = n_ids

    def upload_panel_dev(self, dev_ptr, n_rows, n_ids):
        self._chk(self.lib.ibdg_upload_panel_dev(self.ctx, dev_ptr, n_rows, n_ids))
        self.n_ids = n_ids

    def upload_sites(self, row_index, n_ref, n_alt, window, f_override=None):
        """row_index None: site s is panel row s."""
        ri = None if row_index is None else np.ascontiguousarray(row_index, dtype=np.uint32)
        nr = np.ascontiguousarray(n_ref, dtype=np.uint8)
        na = np.ascontiguousarray(n_alt, dtype=np.uint8)
        fo = None if f_override is None else np.ascontiguousarray(f_override, dtype=np.float64)
        assert len(nr) == len(na) and (ri is None or len(ri) == len(nr))
        self._chk(self.lib.ibdg_upload_sites(self.ctx, None if ri is None else ri.ctypes.data, nr.ctypes.data,
                                             na.ctypes.data, None if fo is None else fo.ctypes.data, len(nr), window))

    def upload_sites_dev(self, row_index_ptr, n_ref_ptr, n_alt_ptr, n_sites, window, f_override=None):
        """The same with the three arrays in device memory (pointers; row_index_ptr may be None/0)."""
        fo = None if f_override is None else np.ascontiguousarray(f_override, dtype=np.float64)
        self._chk(self.lib.ibdg_upload_sites_dev(self.ctx, row_index_ptr or None, n_ref_ptr, n_alt_ptr,
                                                 None if fo is None else fo.ctypes.data, n_sites, window))

    def upload_ms(self):
        """Clocks of the last upload of sites (ms): copies, device preparation, whole call."""
        out = (C.c_float * 3)()
        self._chk(self.lib.ibdg_upload_ms(self.ctx, out))
        return dict(h2d=out[0], device_prep=out[1], call=out[2])

    @property
    def n_sites(self):
        return self.lib.ibdg_num_sites(self.ctx)

    @property
    def n_windows(self):
        return self.lib.ibdg_num_windows(self.ctx)

    def windows(self):
        n = self.n_windows
        first = np.zeros(n, dtype=np.uint32)
        last = np.zeros(n, dtype=np.uint32)
        ncov = np.zeros(n, dtype=np.uint32)
        self._chk(self.lib.ibdg_get_windows(self.ctx, first.ctypes.data, last.ctypes.data, ncov.ctypes.data))
        return first, last, ncov

    def run(self, targets, ld=True, bg_count=None, pu_id=-1):
        t = np.ascontiguousarray(targets, dtype=np.uint32)
        bg = None if bg_count is None else np.ascontiguousarray(bg_count, dtype=np.uint8)
        if bg is not None:
            assert len(bg) == self.n_ids
        self._chk(self.lib.ibdg_run(self.ctx, t.ctypes.data, len(t), None if bg is None else bg.ctypes.data,
                                    int(pu_id), int(bool(ld))))
        self.n_targets = len(t)

    def site_af(self):
        out = np.empty(self.n_sites, dtype=np.float64)
        self._chk(self.lib.ibdg_get_site_af(self.ctx, out.ctypes.data))
        return out

    def site_ll(self, t=0, out=None):
        if out is None:
            out = np.empty((self.n_sites, 3), dtype=np.float64)
        assert out.dtype == np.float64 and out.size >= self.n_sites * 3 and out.flags.c_contiguous
        self._chk(self.lib.ibdg_get_site_ll(self.ctx, t, out.ctypes.data))
        return out

    def window_ll(self, t=0, out=None):
        if out is None:
            out = np.empty((self.n_windows, 3), dtype=np.float64)
        assert out.dtype == np.float64 and out.size >= self.n_windows * 3 and out.flags.c_contiguous
        self._chk(self.lib.ibdg_get_window_ll(self.ctx, t, out.ctypes.data))
        return out

    def window_ll_all(self, n_targets, out=None):
        """The window tables of all `n_targets` comparison individuals of the last run in one copy: [n_targets][n_windows][3]."""
        have = self.lib.ibdg_num_targets(self.ctx)
        if have != n_targets:       # (the C call copies what the LAST RUN produced, whatever the caller's buffer holds)
            raise EngineError(f"window_ll_all({n_targets}): the last run had {have} comparison individuals")
        if out is None:
            out = np.empty((n_targets, self.n_windows, 3), dtype=np.float64)
        assert out.dtype == np.float64 and out.size >= n_targets * self.n_windows * 3 and out.flags.c_contiguous
        self._chk(self.lib.ibdg_get_window_ll_all(self.ctx, out.ctypes.data))
        return out

    def alt_counts(self, first, n):
        out = np.empty(n, dtype=np.uint32)
        self._chk(self.lib.ibdg_get_alt_counts(self.ctx, first, n, out.ctypes.data))
        return out

    def last_run_ms(self):
        out = (C.c_float * 5)()
        self._chk(self.lib.ibdg_last_run_ms(self.ctx, out))
        return dict(total=out[0], alt_count=out[1], rows=out[2], ld=out[3])

    def run_ms(self, back):
        """Device times of the run `back` calls ago (0 = last); waits for the stream."""
        out = (C.c_float * 5)()
        self._chk(self.lib.ibdg_run_ms(self.ctx, back, out))
        return dict(total=out[0], alt_count=out[1], rows=out[2], ld=out[3])

    def run_kernel_ms(self, back=0):
        """Duration of the dominant --LD kernel of the run `back` calls ago (its own dispatch times)."""
        out = C.c_float()
        self._chk(self.lib.ibdg_run_kernel_ms(self.ctx, back, C.byref(out)))
        return out.value

    def set_background_order(self, ids):
        """Background list in the reference's order (reference-order mode); None/empty clears it."""
        a = np.ascontiguousarray([] if ids is None else ids, dtype=np.uint32)
        self._chk(self.lib.ibdg_set_background_order(self.ctx, a.ctypes.data if len(a) else None, len(a)))

    def last_ld_variant(self):
        return self.lib.ibdg_last_ld_variant(self.ctx)

    def last_count_unit(self):
        """2 = the last run's single-individual launches took the sums of a haplotype word on the matrix cores, 1 = by
        (mask, count) pairs, 0 = no such launch (ibdg_last_count_unit)."""
        return self.lib.ibdg_last_count_unit(self.ctx)

    def ld_layout(self):
        """0 none, 1 the panel's own tiles, 2 the compacted tiles of the site list (its rows with reads back to back)."""
        return self.lib.ibdg_ld_layout(self.ctx)

    def sync(self):
        self._chk(self.lib.ibdg_sync(self.ctx))
