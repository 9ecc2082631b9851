"""libfm loader with the reference's surface (drop-in for /root/reference/LoadData.py:15-112).

Same constructor, same attributes (``features``, ``features_M``, ``Train_data``, ``Validation_data``,
``Test_data`` as ``{'X': list of id lists, 'Y': list of floats}``), same console prints, same quirks:

* the dictionary key is the WHOLE token ``"idx:val"``; the value part is never used as a number
  (LoadData.py:49-53, :99);
* ids are handed out in first-appearance order over train, then TEST, then validation
  (LoadData.py:35-39);
* ``features_M`` is the number of distinct tokens, not max-id+1 (LoadData.py:42);
* rows are re-ordered by ``np.argsort`` of their length (LoadData.py:109; not a stable sort, so the
  order among equal-length rows is whatever numpy gives - parity is on the multiset of rows);
* ``log_loss`` maps labels to {0,1} with ``> 0`` (LoadData.py:93-97).

Unlike the reference each file is tokenised once (the reference reads every file twice), and
``packed()`` hands the splits out as dense int32/float32 arrays for the device-resident pipeline.
With the native reader the parsed splits and the token arena are also kept in a binary cache next to the data
(``<path><dataset>/.<dataset>.cffm_cache.npz``, keyed by size and mtime of the three files): a second start skips
the text altogether, and the reference-shaped lists of a split are only built when somebody asks for them
(``data.Train_data['X']``) - the training loop itself works from the packed arrays.
When ``cffm_amd/lib/libcffm_libfm.so`` is present (``make``) the tokenising and the token -> id map run in C++
(mmap + open-addressing hash, ``csrc_host/libfm_reader.cpp``) - same map, same rows; end to end ~2x on the frappe
files because building the reference-shaped list-of-lists dominates.  The pure-Python path stays as the fallback and
as the cross-check in the tests (``native=False``).
"""
import ctypes as C
import os

import numpy as np

# CFFM_HOST_LIB_DIR: another build of the two host shims (the sanitizer build of `make asan`)
_NATIVE_PATH = os.path.join(os.environ.get('CFFM_HOST_LIB_DIR') or
                            os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib'), 'libcffm_libfm.so')
_native = None


def _load_native():
    global _native
    if _native is None and os.path.exists(_NATIVE_PATH):
        lib = C.CDLL(_NATIVE_PATH)
        lib.libfm_open.restype = C.c_void_p
        lib.libfm_close.argtypes = [C.c_void_p]
        lib.libfm_read_file.argtypes = [C.c_void_p, C.c_char_p]
        for name in ('libfm_num_features', 'libfm_arena_bytes'):
            getattr(lib, name).restype = C.c_int64
            getattr(lib, name).argtypes = [C.c_void_p]
        for name in ('libfm_features_after', 'libfm_rows', 'libfm_nnz'):
            getattr(lib, name).restype = C.c_int64
            getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
        lib.libfm_copy_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.libfm_copy_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _native = lib
    return _native


CACHE_VERSION = 1


class _Split(dict):
    """``{'X': list of id lists, 'Y': list of floats}`` (LoadData.py:105-112) whose two lists are built from the packed
    arrays on first access.  Behaves as the plain dict the reference hands out: both keys are always reported, items
    can be re-bound (CFFM.py:183 does), and re-binding drops the packed copy so that ``LoadData.packed`` re-reads the
    lists."""

    def __init__(self, ids2d, y):
        dict.__init__(self)
        self._arrays = (ids2d, y)

    def __missing__(self, key):
        if key not in ('X', 'Y') or self._arrays is None:
            raise KeyError(key)
        ids2d, y = self._arrays
        v = ids2d.tolist() if key == 'X' else y.tolist()
        dict.__setitem__(self, key, v)
        return v

    def __setitem__(self, key, value):
        if self._arrays is not None:                 # materialise the other list first, then forget the arrays
            for k in ('X', 'Y'):
                if k != key and not dict.__contains__(self, k):
                    self.__missing__(k)
            self._arrays = None
        dict.__setitem__(self, key, value)

    def __contains__(self, key):
        return key in ('X', 'Y') or dict.__contains__(self, key)

    def keys(self):
        return ['Y', 'X']

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return 2

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def get(self, key, default=None):
        return self[key] if key in self else default


class LoadData(object):
    # Three files are needed in the path: <path><dataset>/<dataset>.{train,test,validation}.libfm
    def __init__(self, path, dataset, loss_type, native=None, cache=True):
        self._use_native = (_load_native() is not None) if native is None else bool(native)
        if self._use_native and _load_native() is None:
            raise RuntimeError('libcffm_libfm.so is not built (run make)')
        self.path = path + dataset + "/"
        self.trainfile = self.path + dataset + ".train.libfm"
        self.testfile = self.path + dataset + ".test.libfm"
        self.validationfile = self.path + dataset + ".validation.libfm"
        self._cache_file = self.path + "." + dataset + ".cffm_cache.npz" if (cache and self._use_native) else None
        self.cache_hit = False
        self._rows = {}
        self._native_splits = {}
        self._features = None
        self._tokens = None
        self.features_M = self.map_features()
        self.Train_data, self.Validation_data, self.Test_data = self.construct_data(loss_type)

    # -- token -> dense id ---------------------------------------------------------------------
    def map_features(self):
        if self._use_native:
            return self._map_features_native()
        self.features = {}
        for fname in (self.trainfile, self.testfile, self.validationfile):
            self.read_features(fname)
            print(len(self.features))
        return len(self.features)

    # -- binary cache of the parsed text ---------------------------------------------------------
    def _cache_key(self):
        key = [CACHE_VERSION]
        for fname in (self.trainfile, self.testfile, self.validationfile):
            st = os.stat(fname)
            key += [int(st.st_size), int(st.st_mtime_ns)]
        return np.asarray(key, dtype=np.int64)

    def _cache_load(self):
        if not self._cache_file or not os.path.exists(self._cache_file):
            return None
        try:
            z = np.load(self._cache_file, allow_pickle=False)
            if not np.array_equal(z['key'], self._cache_key()):
                return None
            for i, fname in enumerate((self.trainfile, self.testfile, self.validationfile)):
                self._native_splits[fname] = (z['y%d' % i], z['ids%d' % i], z['off%d' % i])
                print(int(z['m_after'][i]))
            self._tokens = (z['arena'].tobytes(), z['tok'])
            self.cache_hit = True
            return int(z['M'])
        except Exception:                          # unreadable / truncated cache: parse the text again
            self._native_splits = {}
            return None

    def _cache_store(self, M, m_after):
        if not self._cache_file:
            return
        try:
            arrays = {'key': self._cache_key(), 'M': np.int64(M), 'm_after': np.asarray(m_after, dtype=np.int64),
                      'arena': np.frombuffer(self._tokens[0], dtype=np.uint8), 'tok': self._tokens[1]}
            for i, fname in enumerate((self.trainfile, self.testfile, self.validationfile)):
                y, ids, off = self._native_splits[fname]
                arrays['y%d' % i], arrays['ids%d' % i], arrays['off%d' % i] = y, ids, off
            tmp = self._cache_file + '.tmp.%d.npz' % os.getpid()
            np.savez(tmp, **arrays)
            os.replace(tmp, self._cache_file)
        except OSError:                            # read-only data directory: run without a cache
            pass

    def _map_features_native(self):
        M = self._cache_load()
        if M is not None:
            return M
        lib = _load_native()
        h = lib.libfm_open()
        m_after = []
        try:
            for fname in (self.trainfile, self.testfile, self.validationfile):     # LoadData.py:35-39 order
                k = lib.libfm_read_file(h, fname.encode())
                if k < 0:
                    raise IOError('cannot read %s' % fname)
                m_after.append(int(lib.libfm_features_after(h, k)))
                print(m_after[-1])
                rows, nnz = int(lib.libfm_rows(h, k)), int(lib.libfm_nnz(h, k))
                y = np.empty(rows, dtype=np.float64)
                ids = np.empty(nnz, dtype=np.int32)
                off = np.empty(rows + 1, dtype=np.int64)
                lib.libfm_copy_split(h, k, y.ctypes.data, ids.ctypes.data, off.ctypes.data)
                self._native_splits[fname] = (y, ids, off)
            M = int(lib.libfm_num_features(h))
            arena = np.empty(max(int(lib.libfm_arena_bytes(h)), 1), dtype=np.uint8)
            tok = np.empty(M + 1, dtype=np.uint32)
            lib.libfm_copy_tokens(h, arena.ctypes.data, tok.ctypes.data)
            self._tokens = (arena.tobytes(), tok)
        finally:
            lib.libfm_close(h)
        self._cache_store(M, m_after)
        return M

    @property
    def features(self):
        """token -> dense id, as the reference keeps it (built lazily from the C++ token arena)."""
        if self._features is None:
            raw, tok = self._tokens
            self._features = {raw[tok[i]:tok[i + 1]].decode(): i for i in range(len(tok) - 1)}
        return self._features

    @features.setter
    def features(self, v):
        self._features = v

    def read_features(self, file):
        feats = self.features
        labels, rows = [], []
        with open(file) as fh:
            for line in fh:
                items = line.strip().split(' ')
                ids = []
                for tok in items[1:]:
                    fid = feats.get(tok)
                    if fid is None:
                        fid = len(feats)
                        feats[tok] = fid
                    ids.append(fid)
                labels.append(items[0])
                rows.append(ids)
        self._rows[file] = (labels, rows)

    # -- splits ----------------------------------------------------------------------------------
    def construct_data(self, loss_type):
        out = []
        for fname, title in ((self.trainfile, "# of training:"),
                             (self.validationfile, "# of validation:"),
                             (self.testfile, "# of test:")):
            if fname in self._native_splits:
                split, rows = self._native_split(fname, loss_type)
                if split is not None:
                    out.append(split)
                    print(title, rows)
                    continue
            X_, Y_, Y_for_logloss = self.read_data(fname)
            out.append(self.construct_dataset(X_, Y_for_logloss if loss_type == 'log_loss' else Y_))
            print(title, len(Y_))
        return tuple(out)

    def _native_split(self, file, loss_type):
        """One split straight from the packed arrays of the native reader / the cache.  Equal-length rows (every CFFM
        data set) become a lazy ``_Split``; ragged rows fall back to eager lists."""
        y, ids, off = self._native_splits[file]
        rows = y.shape[0]
        lens = np.diff(off)
        if rows and (lens == lens[0]).all():
            Y = (y > 0).astype(np.float64) if loss_type == 'log_loss' else y        # LoadData.py:93-97
            order = np.argsort(lens)                                                   # LoadData.py:109
            return _Split(np.ascontiguousarray(ids.reshape(rows, int(lens[0]))[order]), np.ascontiguousarray(Y[order])), rows
        self._rows[file] = (y, [ids[off[i]:off[i + 1]].tolist() for i in range(rows)])
        return None, rows

    def read_data(self, file):
        if file not in self._rows:
            self.read_features(file)
        labels, rows = self._rows[file]
        Y_ = labels.tolist() if isinstance(labels, np.ndarray) else [1.0 * float(t) for t in labels]
        Y_for_logloss = [1.0 if v > 0 else 0.0 for v in Y_]
        return rows, Y_, Y_for_logloss

    def construct_dataset(self, X_, Y_):
        order = np.argsort([len(r) for r in X_])
        return {'Y': [Y_[i] for i in order], 'X': [X_[i] for i in order]}

    def truncate_features(self):
        """Cut every row to the shortest training row (LoadData.py:114-128; CFFM never calls it)."""
        n = min(len(r) for r in self.Train_data['X'])
        for d in (self.Train_data, self.Validation_data, self.Test_data):
            d['X'] = [r[:n] for r in d['X']]
        return n

    # -- extension: dense arrays for the device pipeline -------------------------------------------
    @staticmethod
    def packed(data):
        """``{'X','Y'}`` -> (int32 [N,F], float32 [N]); requires equal-length rows."""
        arrays = getattr(data, '_arrays', None)
        if arrays is not None:                      # lists never built / never re-bound: the packed copy is current
            return np.ascontiguousarray(arrays[0], dtype=np.int32), arrays[1].astype(np.float32)
        X = np.asarray(data['X'], dtype=np.int32)
        if X.ndim != 2:
            raise ValueError('rows have different lengths; the CFFM graph needs num_field ids per row')
        return np.ascontiguousarray(X), np.asarray(data['Y'], dtype=np.float32)
