"""Host half of the handle cache (include/locrec.h, "Handle cache"; csrc/cache.hip).

The reference's mains construct a new recommender from freshly read DataFrames for EVERY request
(knn/KnnRecommenderMain.scala:53-67, stochastic/StochasticRecommenderMain.scala:53-62) and close
nothing.  Behind those unchanged mains the constructor must therefore be cheap the second time: it
derives a key from what its frames ARE and asks the library's process-wide cache for the device
handle; only a miss collects, uploads and builds.  The weights, K, epsilon and maxIterations are
arguments of every request call of the C ABI and deliberately NOT part of the key.

Key of a frame, in this order (the Scala classes do the first; pandas needs the other two):
  1. files   - a frame read from disk carries `attrs["inputFiles"]` (mains.read_parquet_frame sets it, the
               stand-in for Spark's `df.inputFiles`): the sorted file list with sizes and modification times.
  2. memo    - the very same frame OBJECT seen before (weak reference, so a recycled id() cannot alias):
               frames are treated as immutable, as Spark DataFrames are.
  3. content - a 128-bit digest of the frame's columns (for rating vectors: size, indices, values of every row).

LOCREC_BACKEND: `gpu` (default) or `spark`.  `spark` selects
the reference's own Spark implementation, which only exists on the JVM side; this Python mirror refuses it
loudly instead of computing anything on the CPU.
"""
import ctypes as C
import hashlib
import os
import threading
import weakref

import numpy as np

from . import _lib as L


def backend():
    b = os.environ.get("LOCREC_BACKEND", "gpu").strip().lower()
    if b not in ("gpu", "spark"):
        raise L.IllegalArgumentException(f"LOCREC_BACKEND must be 'gpu' or 'spark': {b}")
    return b


def require_gpu_backend(what):
    if backend() == "spark":
        raise L.LocrecRuntimeError(
            f"LOCREC_BACKEND=spark: {what} would run the reference's Spark implementation, which exists only "
            "behind the Scala classes (scala/.../LocrecBackend.scala); this host mirror has no CPU path")


# ---- keys -------------------------------------------------------------------------------------------

_memo = {}            # id(frame) -> (weakref, key)
_memo_lock = threading.Lock()


def files_key(paths):
    """Spark: df.inputFiles.sorted + length + modification time of every file."""
    parts = []
    for p in sorted(paths):
        for q in ([p] if os.path.isfile(p) else sorted(
                os.path.join(dp, f) for dp, _, fs in os.walk(p) for f in fs if not f.startswith(("_", ".")))):
            st = os.stat(q)
            parts.append(f"{q}:{st.st_size}:{st.st_mtime_ns}")
    return "files:" + hashlib.blake2b("|".join(parts).encode(), digest_size=16).hexdigest()


def _digest_column(h, col):
    a = np.asarray(col)
    if a.dtype == object:  # rating_vector: SparseVector objects
        for v in a:
            h.update(np.int64(v.size).tobytes())
            h.update(np.ascontiguousarray(v.indices).view(np.uint8))
            h.update(np.ascontiguousarray(v.values).view(np.uint8))
            h.update(b"/")
    else:
        h.update(str(a.dtype).encode())
        h.update(np.ascontiguousarray(a).view(np.uint8))


def content_key(frame, columns):
    h = hashlib.blake2b(digest_size=16)
    for c in columns:
        h.update(c.encode() + b"=")
        _digest_column(h, frame[c])
    return "content:" + h.hexdigest()


def frame_key(frame, columns):
    files = getattr(frame, "attrs", {}).get("inputFiles") if hasattr(frame, "attrs") else None
    if files:
        return files_key(files)
    with _memo_lock:
        m = _memo.get(id(frame))
        if m is not None and m[0]() is frame:
            return m[1]
    key = content_key(frame, columns)
    try:
        ref = weakref.ref(frame, lambda _, i=id(frame): _memo.pop(i, None))
        with _memo_lock:
            _memo[id(frame)] = (ref, key)
    except TypeError:
        pass  # a plain dict of columns: no weak references, content key every time
    return key


# ---- handles ----------------------------------------------------------------------------------------

_locks = {}           # handle address -> lock shared by every host object that references the handle
_locks_guard = threading.Lock()


def handle_lock(h):
    with _locks_guard:
        return _locks.setdefault(int(h.value if hasattr(h, "value") else h), threading.RLock())


def acquire(kind, key):
    """The cached handle for key (a reference is taken) or None."""
    h = C.c_void_p()
    L.check(L.lib().locrec_cache_acquire(kind, key.encode(), C.byref(h)))
    return h if h.value else None


def publish(kind, key, handle, device_bytes):
    """Hand a freshly created handle to the cache; returns the handle to use (a reference is taken)."""
    out = C.c_void_p()
    L.check(L.lib().locrec_cache_publish(kind, key.encode(), handle, int(device_bytes), C.byref(out)))
    return out


def release(kind, handle):
    if handle is not None and getattr(handle, "value", handle):
        L.lib().locrec_cache_release(kind, handle)


def set_limits(max_device_bytes=-1, max_entries=-1):
    L.check(L.lib().locrec_cache_set_limits(int(max_device_bytes), int(max_entries)))


def clear():
    L.check(L.lib().locrec_cache_clear())
