"""Proof stream / Fiat-Shamir -- mirror of the reference's `ip.py` (/root/reference/code/ip.py:4-30).

`objects` is a plain Python list, as in the reference.  The byte stream the reference obtains from
`pickle.dumps(self.objects)` is produced by the native transcript code (csrc/refpickle.hpp) from an equivalent
object graph: elements of this package's classes are emitted exactly as the reference's `algebra.*`,
`univariate.*`, `extension_field.*` instances would be, shared Python objects stay shared (pickle memoises by
identity), and SHAKE256 is applied to those bytes.
"""
import ctypes
import io
import pickle

from . import _lib
from .algebra import BaseField, BaseFieldElement
from .extension_field import ExtensionField, ExtensionFieldElement
from .univariate import Polynomial

_u64 = ctypes.c_uint64


class NativeTranscript:
    """owns a bfs_ps_* handle and the two-way mapping between Python objects and native object handles."""

    _dump_buffer = None   # dumps_handle's scratch, per instance (verifiers may run in several threads)
    loaded = False        # True: made by from_bytes -- the native side was read from the pickle, no Python object maps to a handle

    def __init__(self, _handle=None):
        self.lib = _lib.load()
        self.handle = self.lib.bfs_ps_new() if _handle is None else _handle
        self._by_id = {}     # id(python object) -> native handle
        self._keep = []      # keeps those python objects alive so ids stay unique
        self._by_handle = {}  # native handle -> python object (identity of objects created natively)
        self._fields = {}    # id(BaseField instance) -> native field id
        self._coefficient_uses = {}   # id(coefficient object) -> number of extension elements holding it (scan())
        self._compact_coefficients = set()   # ids of coefficient objects of elements written in compact form
        self.xfield = None

    def __del__(self):
        try:
            self.lib.bfs_ps_free(self.handle)
        except Exception:
            pass

    @classmethod
    def from_bytes(cls, data):
        """the native stream of a serialised proof (bfs_ps_loads: read natively, accepted only if it serialises back to `data`), or
        None when the bytes hold something the native reader does not take -- the caller then builds the stream from Python objects"""
        data = bytes(data)
        handle = _lib.load().bfs_ps_loads(data, len(data))
        if not handle:
            return None
        t = cls(_handle=handle)
        t.loaded = True
        return t

    def dumps_handle(self, h):
        """pickle.dumps of the object behind native handle h, on its own"""
        lib = self.lib
        n = ctypes.c_size_t()
        buf = self._dump_buffer                 # one call when the pickle fits (a row of 26 elements is < 4 KiB)
        if buf is None:
            buf = self._dump_buffer = ctypes.create_string_buffer(8192)
        _lib.check(lib.bfs_ps_obj_dumps(self.handle, h, buf, len(buf), ctypes.byref(n)))
        if n.value > len(buf):
            buf = self._dump_buffer = ctypes.create_string_buffer(2 * n.value)
            _lib.check(lib.bfs_ps_obj_dumps(self.handle, h, buf, len(buf), ctypes.byref(n)))
        return buf.raw[:n.value] if n.value * 4 > len(buf) else ctypes.string_at(buf, n.value)

    def scan(self, objs):
        """note which BaseFieldElement objects are coefficients of more than one extension element: those elements must be
        built over explicit coefficient objects so that the second occurrence becomes a back-reference, as in pickle."""
        # (a loop with its own stack: a nested function that calls itself is a reference cycle -- function, closure cell, function --
        # that also holds `self`, so every call kept a transcript and its native object graph alive until the cyclic collector ran)
        seen, uses = set(), self._coefficient_uses
        stack = list(objs)[::-1]
        while stack:
            o = stack.pop()
            if isinstance(o, ExtensionFieldElement):
                if id(o) not in seen:
                    seen.add(id(o))
                    for c in o.polynomial.coefficients:
                        uses[id(c)] = uses.get(id(c), 0) + 1
            elif isinstance(o, (list, tuple)):
                if id(o) not in seen:
                    seen.add(id(o))
                    stack.extend(reversed(o))

    def _field_id(self, field):
        """BaseField instances are distinguished by identity, like pickle does: 1 = the one inside the xfield's modulus,
        0 = the first other instance met, 2.. = further ones."""
        if self.xfield is not None and field is self.xfield.modulus.coefficients[0].field:
            return 1
        key = id(field)
        if key not in self._fields:
            self._fields[key] = 0 if not self._fields else len(self._fields) + 1
            self._keep.append(field)
        return self._fields[key]

    # ---- python -> native
    def to_native(self, obj):
        key = id(obj)
        if key in self._by_id:
            return self._by_id[key]
        lib, h = self.lib, self.handle
        if isinstance(obj, (bytes, bytearray)):
            n = lib.bfs_ps_obj_bytes(h, bytes(obj), len(obj))
        elif isinstance(obj, ExtensionFieldElement):
            if self.xfield is None:
                self.xfield = obj.field
            coeffs = obj.polynomial.coefficients
            internal = self.xfield.modulus.coefficients[0].field
            plain = all(c.field is internal and id(c) not in self._by_id and self._coefficient_uses.get(id(c), 0) < 2 for c in coeffs)
            if plain and not getattr(obj, "shares_coefficients", False):
                n = lib.bfs_ps_obj_xfe(h, (_u64 * 3)(*obj.limbs()))
                self._compact_coefficients.update(id(c) for c in coeffs)
            else:
                # coefficient objects that are shared with another element, or that point at a foreign BaseField instance
                handles = [self.to_native(c) for c in coeffs]
                n = lib.bfs_ps_obj_xfe_from(h, (_u64 * len(handles))(*handles), len(handles))
        elif isinstance(obj, BaseFieldElement):
            n = lib.bfs_ps_obj_bfe(h, obj.value, self._field_id(obj.field))
        elif isinstance(obj, bool):
            raise TypeError("bool objects are not supported in the native transcript")
        elif isinstance(obj, int):
            if not 0 <= obj < 1 << 64:
                raise TypeError("only integers in [0, 2^64) are supported in the native transcript")
            return lib.bfs_ps_obj_int(h, obj)            # ints are never memoised by pickle
        elif isinstance(obj, (list, tuple)):
            items = [self.to_native(x) for x in obj]
            arr = (_u64 * len(items))(*items)
            n = (lib.bfs_ps_obj_list if isinstance(obj, list) else lib.bfs_ps_obj_tuple)(h, arr, len(items))
        else:
            raise TypeError("cannot put a %s into the native proof stream" % type(obj).__name__)
        if n == 0:
            _lib.check(6)
        self._by_id[key] = n
        self._keep.append(obj)
        self._by_handle[n] = obj
        return n

    # ---- native -> python
    def to_python(self, n, xfield):
        if n in self._by_handle:
            return self._by_handle[n]
        lib, h = self.lib, self.handle
        kind = lib.bfs_ps_obj_kind(h, n)
        if kind == 0:
            buf = ctypes.create_string_buffer(max(lib.bfs_ps_obj_len(h, n), 1))
            _lib.check(lib.bfs_ps_obj_get_bytes(h, n, buf, len(buf)))
            obj = buf.raw[:lib.bfs_ps_obj_len(h, n)]
        elif kind in (1, 100, 101):
            limbs = (_u64 * 3)()
            _lib.check(lib.bfs_ps_obj_get_limbs(h, n, limbs))
            if kind == 1:
                return int(limbs[0])
            obj = xfield.from_limbs(list(limbs)) if kind == 100 else BaseFieldElement(int(limbs[0]), xfield.modulus.coefficients[0].field)
        elif kind in (3, 4):
            items = [self.to_python(lib.bfs_ps_obj_item(h, n, i), xfield) for i in range(lib.bfs_ps_obj_len(h, n))]
            obj = items if kind == 3 else tuple(items)
        else:
            raise RuntimeError("unexpected native object kind %d" % kind)
        self._by_handle[n] = obj
        self._by_id[id(obj)] = n
        self._keep.append(obj)
        return obj

    def push(self, obj):
        _lib.check(self.lib.bfs_ps_push(self.handle, self.to_native(obj)))

    _serialize_buffer = None      # per instance, grows to the largest stream seen (one native call per serialize() when it fits)

    def serialize(self, count=None):
        lib = self.lib
        count = (1 << 62) if count is None else count
        n = ctypes.c_size_t()
        buf = self._serialize_buffer
        if buf is None:
            buf = self._serialize_buffer = ctypes.create_string_buffer(1 << 16)
        _lib.check(lib.bfs_ps_serialize(self.handle, count, buf, len(buf), ctypes.byref(n)))
        if n.value > len(buf):
            buf = self._serialize_buffer = ctypes.create_string_buffer(2 * n.value)
            _lib.check(lib.bfs_ps_serialize(self.handle, count, buf, len(buf), ctypes.byref(n)))
        return ctypes.string_at(buf, n.value)

    def dumps(self, obj):
        """pickle.dumps(obj) for one object on its own (leaf preimages)."""
        lib = self.lib
        h = self.to_native(obj)
        n = ctypes.c_size_t()
        _lib.check(lib.bfs_ps_obj_dumps(self.handle, h, None, 0, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(max(n.value, 1))
        _lib.check(lib.bfs_ps_obj_dumps(self.handle, h, buf, n.value, ctypes.byref(n)))
        return buf.raw[:n.value]

    def fiat_shamir(self, count, num_bytes):
        out = ctypes.create_string_buffer(num_bytes)
        _lib.check(self.lib.bfs_ps_fiat_shamir(self.handle, (1 << 62) if count is None else count, out, num_bytes))
        return out.raw

    def num_objects(self):
        return self.lib.bfs_ps_num_objects(self.handle)


def _find_xfield(obj):
    """the ExtensionField of the first extension element inside obj (None if there is none)."""
    if isinstance(obj, ExtensionFieldElement):
        return obj.field
    if isinstance(obj, (list, tuple)):
        for x in obj:
            f = _find_xfield(x)
            if f is not None:
                return f
    return None


def reference_pickle(obj):
    """bytes of the reference's `pickle.dumps(obj)` for an object made of this package's element classes,
    bytes, ints, lists and tuples."""
    return NativeTranscript().dumps(obj)


class ProofStream:
    def __init__(self):
        self._objects = []
        self._pending = None      # (transcript, first, last, field): objects native code appended, not yet turned into Python objects
        self.read_index = 0

    @property
    def objects(self):
        """the list of pushed objects, as in the reference.  Objects appended by native code (Fri.prove writes ~10^3 of them)
        become Python objects only when somebody looks: the prover itself serialises and hashes through the native transcript."""
        if self._pending is not None:
            transcript, first, last, field = self._pending
            self._pending = None
            new = [transcript.to_python(transcript.lib.bfs_ps_object_at(transcript.handle, i), field) for i in range(first, last)]
            self._objects += new
            if getattr(self, "_cached", None) is transcript:
                self._cached_ids += [id(o) for o in new]
        return self._objects

    @objects.setter
    def objects(self, value):
        self._pending = None
        self._objects = value

    def push(self, obj):
        self.objects += [obj]

    def pull(self):
        assert self.read_index < len(self.objects), "ProofStream: cannot pull object; queue empty."
        obj = self.objects[self.read_index]
        self.read_index += 1
        return obj

    def _native(self, count=None):
        """native transcript holding objects[:count] (all when count is None).  The full-stream transcript is kept and
        extended as objects are appended (a proof asks for Fiat-Shamir randomness several times, each time over everything
        pushed so far); it is rebuilt from scratch whenever an earlier choice could have been wrong: the list was edited
        in place, an extension field shows up after bare base elements, or a coefficient object turns out to be shared
        with an element that was already written in compact form."""
        if count is None and self._pending is not None and getattr(self, "_cached", None) is self._pending[0]:
            return self._cached                   # the cached transcript is ahead of the Python list, and complete
        if count is not None and count != len(self.objects):
            return self._build(self.objects[:count])
        t = getattr(self, "_cached", None)
        objs = self.objects
        if t is not None:
            k = len(self._cached_ids)
            stale = k > len(objs) or list(map(id, objs[:k])) != self._cached_ids      # (one C-level pass: the verifier asks ~20 times per proof)
            if not stale and k < len(objs) and t.loaded:
                stale = True          # read natively from bytes: it has no map from Python objects to handles, so it cannot take more
            if not stale and k < len(objs):
                new = objs[k:]
                if t.xfield is None and _find_xfield(new) is not None:
                    stale = True
                else:
                    before = dict(t._coefficient_uses)
                    t.scan(new)
                    stale = any(uses >= 2 and before.get(c, 0) == 1 and c in t._compact_coefficients
                                for c, uses in t._coefficient_uses.items())
                    if not stale:
                        for o in new:
                            t.push(o)
                        self._cached_ids += [id(o) for o in new]
            if not stale:
                return t
        t = self._build(objs)
        self._cached, self._cached_ids = t, [id(o) for o in objs]
        return t

    @staticmethod
    def _build(objs):
        t = NativeTranscript()
        t.xfield = _find_xfield(objs)     # decides which BaseField instance a bare BaseFieldElement refers to
        t.scan(objs)
        for o in objs:
            t.push(o)
        return t

    def _adopt(self, transcript, new_objects):
        """objects appended by native code (Fri.prove) to the cached transcript: keep list and cache in step"""
        self.objects += new_objects
        if getattr(self, "_cached", None) is transcript:
            self._cached_ids += [id(o) for o in new_objects]

    def _adopt_lazy(self, transcript, first, last, field):
        """like _adopt, for objects first..last-1 of the cached transcript, without building them"""
        if self._pending is not None and self._pending[0] is transcript and self._pending[2] == first and getattr(self, "_cached", None) is transcript:
            self._pending = (transcript, self._pending[1], last, field)          # native code went on appending: one longer range
        elif getattr(self, "_cached", None) is not transcript or self._pending is not None:
            self._adopt(transcript, [transcript.to_python(transcript.lib.bfs_ps_object_at(transcript.handle, i), field) for i in range(first, last)])
        else:
            self._pending = (transcript, first, last, field)

    def serialize(self):
        return self._native().serialize()

    def prover_fiat_shamir(self, num_bytes=32):
        return self._native().fiat_shamir(None, num_bytes)

    def verifier_fiat_shamir(self, num_bytes=32):
        # SHAKE256 of pickle.dumps(objects[:read_index]) (ip.py:29-30).  The pickle of a prefix of the list is what the
        # (cached) transcript of the whole list holds for its first read_index objects -- memo indices and frame cuts only
        # depend on what came before -- so nothing is re-encoded per call.
        return self._native().fiat_shamir(self.read_index, num_bytes)

    def deserialize(self, bb):
        """ip.py:27-30.  The Python objects come from CPython's unpickler (into this package's classes); the verifier's Fiat-Shamir
        calls and leaf pickles need the BYTES of prefixes and of single objects, and those come from a native stream read from the
        same bytes (bfs_ps_loads) instead of from walking the Python objects again -- that walk was half of verify()'s time."""
        ps = ProofStream()
        ps.objects = _ReferenceUnpickler(io.BytesIO(bb)).load()
        t = NativeTranscript.from_bytes(bb) if isinstance(ps._objects, list) else None
        if t is not None and t.num_objects() == len(ps._objects):
            t.xfield = _find_xfield(ps._objects)
            ps._cached, ps._cached_ids = t, [id(o) for o in ps._objects]
            # where a pulled object (or an item of a pulled tuple: FRI's (a, b, c) leaves) sits in the native stream
            handles = {}
            for k, o in enumerate(ps._objects):
                handles[id(o)] = (k + 1, None, o)
                if isinstance(o, tuple):
                    for i, c in enumerate(o):
                        handles.setdefault(id(c), (k + 1, i, c))
            ps._handles = handles
            # Fiat-Shamir ahead of time: the verifiers ask right before / after a top-level digest (a Merkle root) and after a codeword;
            # those prefixes go to the helper threads now (a position that was not foreseen is simply hashed when it is asked for)
            at = set()
            for k, o in enumerate(ps._objects):
                if isinstance(o, (bytes, bytearray)) and len(o) == 64:
                    at.update((k, k + 1))
                elif isinstance(o, list) and o and isinstance(o[0], ExtensionFieldElement):
                    at.add(k + 1)
            at.discard(0)
            if at:
                counts = (ctypes.c_size_t * len(at))(*sorted(at, reverse=True))          # the long prefixes first
                t.lib.bfs_ps_prefetch_fiat_shamir(t.handle, counts, len(at), 32)
        return ps

    def _handle_of(self, obj):
        """native handle of an object of a deserialised stream (a pushed object or an item of a pushed tuple), by identity; 0: not one"""
        entry = getattr(self, "_handles", {}).get(id(obj))
        t = getattr(self, "_cached", None)
        if entry is None or t is None or not t.loaded or entry[2] is not obj:
            return 0
        h, item, _ = entry
        if item is not None:
            h = t.lib.bfs_ps_obj_item(t.handle, h, item)
        return h

    def native_path_check(self, root, index, salt, path, element):
        """Merkle.verify / SaltedMerkle.verify (merkle.py:54-63, salted_merkle.py:55-68) for objects of this deserialised stream in one
        native call (bfs_ps_merkle_verify); None when one of the objects is not from the stream (the caller then hashes in Python)"""
        he, hp = self._handle_of(element), self._handle_of(path)
        hs = self._handle_of(salt) if salt is not None else 0
        if not he or not hp or (salt is not None and not hs) or not isinstance(root, (bytes, bytearray)) or index < 0 or index >> 64:
            return None
        t = self._cached
        ok = ctypes.c_int(0)
        _lib.check(t.lib.bfs_ps_merkle_verify(t.handle, he, hs, hp, index, bytes(root), len(root), ctypes.byref(ok)))
        return bool(ok.value)

    def pickle_of(self, obj):
        """pickle.dumps(obj) for an object of a deserialised stream (by identity), from the native stream; None when it is not one"""
        entry = getattr(self, "_handles", {}).get(id(obj))
        t = getattr(self, "_cached", None)
        if entry is None or t is None or not t.loaded or entry[2] is not obj:
            return None
        h, item, _ = entry
        if item is not None:
            h = t.lib.bfs_ps_obj_item(t.handle, h, item)
            if not h:
                return None
        return t.dumps_handle(h)


class _ReferenceUnpickler(pickle.Unpickler):
    """reads a stream written by the reference (or by serialize()) into this package's classes."""
    _CLASSES = {("algebra", "BaseField"): BaseField, ("algebra", "BaseFieldElement"): BaseFieldElement,
                ("univariate", "Polynomial"): Polynomial, ("extension_field", "ExtensionField"): ExtensionField,
                ("extension_field", "ExtensionFieldElement"): ExtensionFieldElement}

    def find_class(self, module, name):
        try:
            return self._CLASSES[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError("refusing to load %s.%s from a proof stream" % (module, name))
