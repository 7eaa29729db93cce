"""Execution-trace tables -- mirror of the reference's `table.py` (/root/reference/code/table.py:8-341) with the heavy
steps on the GPU:

  interpolate_columns + lde / ldex   (:112-148)  INTT over the omicron subgroup, rank-one randomizer correction
                                                 (bfs_poly_randomize), coset NTT onto the FRI domain -- per table one
                                                 batched call over all columns, codewords stay in HBM
  extend                             (per table)  the running products / evaluations as scan specs (`_scans`): `extend_device`
                                                 runs them as prefix scans on the trace columns in HBM (bfs_xfe_scan_device),
                                                 `extend` on the host primitive (bfs_xfe_scan) for the reference's call surface
  all_quotients                      (:148-281)  one kernel per table (bfs_air_quotients, constraints from air.py); the prover
                                                 itself uses `combine_into` (bfs_air_combine: quotients folded into the
                                                 non-linear combination, never written) with `zerofier_inverses`
  *_quotient_degree_bounds           (:170-173, 238-247, 300-304)  exact symbolic expansion on the host (air.expand), the generic
                                                 surviving-monomial pattern cached for sampled challenges

Base columns live on the host as uint64 arrays (in pinned staging memory once padded); extension columns exist on the host only
when `extend` (not `extend_device`) made them.
"""
import ctypes
import itertools
from os import urandom          # module-level on purpose: tests patch `table.urandom` for determinism
from .randomness import source as random_source

import numpy as np

from . import _lib, air
from .algebra import BaseFieldElement
from .arrays import raw_ntt
from .device import DeviceBuffer, current_stream

_ARRAY_GENERATION = itertools.count(1)      # keys of Table's per-matrix caches: never reused

P = air.P
_u64 = ctypes.c_uint64


def _val(x):
    return x.value if hasattr(x, "value") else int(x)


_INVERSES = {}


def _inv(v):
    """1 / v mod p, remembered: a proof asks ~30 times for the inverses of the same five subgroup generators and heights, and a
    modular exponentiation of Python integers is ~5 us"""
    r = _INVERSES.get(v)
    if r is None:
        if len(_INVERSES) > 4096:
            _INVERSES.clear()
        r = _INVERSES[v] = pow(v, P - 2, P)
    return r


def sample_base(byte_array):
    """BaseField.sample (algebra.py:138-142)"""
    return int.from_bytes(bytes(byte_array), "big") % P


def sample_ext_many(data, count, chunk):
    """`count` extension elements from count * 3 * chunk bytes (chunk <= 15): ExtensionField.sample applied to consecutive
    3*chunk-byte strings, vectorised.  Returns a uint64 array of shape (3, count) (limb planes)."""
    raw = np.frombuffer(data, dtype=np.uint8).reshape(count * 3, chunk)
    p = np.uint64(P)
    eps = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        if chunk <= 8:
            padded = np.zeros((count * 3, 8), dtype=np.uint8)
            padded[:, 8 - chunk:] = raw
            out = padded.view(">u8").reshape(-1).astype(np.uint64)
            out = np.where(out >= p, out - p, out)
        elif chunk == 9:
            # value = top * 2^64 + low, 2^64 = 2^32 - 1 (mod p): one multiply-add instead of a Horner loop over the bytes
            low = np.ascontiguousarray(raw[:, 1:]).view(">u8").reshape(-1).astype(np.uint64)
            low = np.where(low >= p, low - p, low)
            out = low + raw[:, 0].astype(np.uint64) * eps            # < p + 2^40: wraps at most once
            out = np.where(out < low, out + eps, out)
            out = np.where(out >= p, out - p, out)
        else:
            out = np.zeros(count * 3, dtype=np.uint64)
            for b in range(chunk):                      # Horner over the big-endian bytes: acc = acc * 256 + byte  (mod p)
                hi = out >> np.uint64(56)               # the 8 bits shifted out are worth hi * 2^64 = hi * (2^32 - 1)
                lo = out << np.uint64(8)
                lo = np.where(lo >= p, lo - p, lo)
                t = lo + hi * eps                       # lo < p, hi * eps < 2^40: at most one wrap
                t = np.where(t < lo, t + eps, t)
                t = np.where(t >= p, t - p, t)
                t2 = t + raw[:, b].astype(np.uint64)
                t2 = np.where(t2 < t, t2 + eps, t2)
                out = np.where(t2 >= p, t2 - p, t2)
    return out.reshape(count, 3).T.copy()


def sample_ext(byte_array):
    """ExtensionField.sample (extension_field.py:100-111): three equal chunks"""
    chunk = len(byte_array) // 3
    return tuple(sample_base(byte_array[i * chunk:(i + 1) * chunk]) for i in range(3))


_POOLS = {}
_THREAD_ROWS = 1 << 14


def _scan_pool():
    if "scan" not in _POOLS:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOLS["scan"] = ThreadPoolExecutor(max_workers=max(2, min(9, os.cpu_count() or 2)), thread_name_prefix="bfs-scan")
    return _POOLS["scan"]


def extend_all(tables, challenges, initials):
    """Table.extend (the host scans) of every table, concurrently (brainfuck_stark.py:186-187 runs them one after the other)"""
    if max(t.height for t in tables) < _THREAD_ROWS:
        for t in tables:
            t.extend(challenges, initials)
        return
    if "tables" not in _POOLS:
        from concurrent.futures import ThreadPoolExecutor
        _POOLS["tables"] = ThreadPoolExecutor(max_workers=5, thread_name_prefix="bfs-extend")
    for future in [_POOLS["tables"].submit(t.extend, challenges, initials) for t in tables]:
        future.result()


def _cache_key(t):
    return t._array_key if t._array_current() else None


def prepare_extension(tables):
    """the challenge-independent part of extend_tables_device -- row masks computed and queued for upload -- done ahead of time; returns
    a token for extend_tables_device (or None when there is nothing to prepare)"""
    lib, stream = _lib.load(), current_stream()
    masks = []
    for t in tables:
        if t.height:
            masks += [np.ascontiguousarray(m, dtype=np.uint8) for m in t._scan_masks() if m is not None]
    if not masks:
        return None
    host = staging_empty(((sum(m.size for m in masks) + 7) // 8,)).view(np.uint8)
    np.concatenate(masks, out=host[:sum(m.size for m in masks)])
    d_masks = DeviceBuffer(host.size // 8)
    _lib.check(lib.bfs_memcpy_h2d(d_masks.ptr, host.ctypes.data, host.size, stream))
    return {"d_masks": d_masks, "host": host, "keys": [_cache_key(t) for t in tables]}


def extend_tables_device(tables, all_challenges, all_initials, prepared=None):
    """Table.extend_device for several tables with ONE upload (all row masks), ONE read-back of the terminals and one gather for
    the values the tables look up afterwards -- a proof has nine scans over five tables, and every separate copy is a round trip."""
    from .device import GatherBatch
    lib, stream = _lib.load(), current_stream()
    plans, masks = [], []
    use_prepared = prepared is not None and prepared["keys"] == [_cache_key(t) for t in tables]
    for t in tables:
        specs = t._scans(all_challenges, all_initials)
        assert len(specs) == t.full_width - t.base_width
        t.ext_columns = None
        t._ext_device = DeviceBuffer(3 * len(specs) * t.height)
        assert t.height == 0 or t._base_device is not None, "extend_device() follows lde()"
        plans.append((t, specs))
        if t.height and not use_prepared:
            masks += [np.ascontiguousarray(sp["mask"], dtype=np.uint8) for sp in specs if sp["mask"] is not None]
    d_masks = None
    if use_prepared:
        d_masks = prepared["d_masks"]           # (same tables, same padded matrices: the masks are already on their way)
    elif masks:
        host = np.concatenate(masks)
        d_masks = DeviceBuffer((host.size + 7) // 8)
        _lib.check(lib.bfs_memcpy_h2d(d_masks.ptr, host.ctypes.data, host.size, stream))
    total = sum(len(specs) for _, specs in plans)
    d_terminals = DeviceBuffer(3 * max(total, 1))
    mask_at, slot, batch_specs = 0, 0, []
    for t, specs in plans:
        h = t.height
        for k, sp in enumerate(specs):
            if h:
                ptrs = [t._base_device.ptr + 8 * c * h for c in sp["cols"]] + [None] * (3 - len(sp["cols"]))
                mask_ptr = None
                if sp["mask"] is not None:
                    mask_ptr = d_masks.ptr + mask_at
                    mask_at += h
                flat = [v for c in sp["constants"] for v in c] + [0] * (12 - 3 * len(sp["constants"]))
                batch_specs.append(_lib.ScanSpec(sp["kind"], 1 if sp["before"] else 0, ptrs[0], ptrs[1], ptrs[2], sp.get("shift1", 0),
                                                 mask_ptr, h, (_u64 * 12)(*flat), (_u64 * 3)(*sp["initial"]),
                                                 t._ext_device.ptr + 8 * 3 * k * h, h, d_terminals.ptr + 24 * slot))
            slot += 1
    if batch_specs:          # every scan of the proof side by side: three launches
        _lib.check(lib.bfs_xfe_scan_device_many((_lib.ScanSpec * len(batch_specs))(*batch_specs), len(batch_specs), stream))
    batch = GatherBatch()
    terminal_ticket = batch.add(d_terminals.ptr, 3 * max(total, 1), 1)
    read_tickets = {(t, k, row): batch.add(t._ext_device.ptr + 8 * (3 * k * t.height + row), 3, t.height)
                    for t, _ in plans if t.height for k, row in t._terminal_reads()}
    batch.run(stream)
    words = [int(v) for v in batch.words(terminal_ticket)]
    slot = 0
    for t, specs in plans:
        if t.height:
            terminals = [tuple(words[3 * (slot + k):3 * (slot + k) + 3]) for k in range(len(specs))]
        else:
            terminals = [tuple(sp["initial"]) for sp in specs]
        slot += len(specs)
        t._after_extend(terminals, all_challenges,
                        lambda k, row, t=t: tuple(int(v) for v in batch.words(read_tickets[(t, k, row)])))


def lde_tables(tables, domain, extension=False):
    """Table.lde (extension=False) or Table.ldex (True) of several tables with ONE transform onto the FRI domain: every table still
    interpolates its own columns (its own subgroup), into a shared coefficient buffer, but the coset evaluation -- the same
    N-point transform for all of them -- runs once over all columns (three launches instead of three per table; on short traces the
    launches are the cost).  Randomizers are drawn table by table, column by column, as the separate calls draw them."""
    lib, stream = _lib.load(), current_stream()
    n = domain.length
    log_n = n.bit_length() - 1
    widths = [3 * (t.full_width - t.base_width) if extension else t.base_width for t in tables]
    total, hmax = sum(widths), max(t.height for t in tables)
    if total == 0 or hmax == 0:
        for t in tables:
            t.ldex(domain) if extension else t.lde(domain)
        return
    randomized = any(t.height and t.num_randomizers for t in tables)
    stride = hmax + 1
    n_in = hmax + 1 if randomized else hmax
    assert n_in <= n, "interpolant does not fit the FRI domain"
    coeffs = DeviceBuffer(total * stride)
    _lib.check(lib.bfs_memset(coeffs.ptr, 0, coeffs.nbytes, stream))
    out = DeviceBuffer(total * n)
    omega, offset = domain.omega.value, domain.offset.value
    at, inputs = 0, []
    for t, w in zip(tables, widths):
        h = t.height
        d_in = None
        if h and w:
            if extension:
                if getattr(t, "_ext_device", None) is not None:      # after extend_device(): already in HBM, (column, limb) planes
                    d_in = t._ext_device
                else:
                    cols = staging_empty((w, h))
                    np.concatenate(t.ext_columns, axis=0, out=cols)
                    d_in = DeviceBuffer.from_numpy(cols.reshape(-1))
            else:
                d_in = DeviceBuffer.from_numpy(t.base_array().reshape(-1))
            rand = None
            if t.num_randomizers:
                if extension:
                    rand = [v for _ in range(w // 3) for v in sample_ext(random_source(urandom)(3 * 8))]
                else:
                    rand = [sample_base(random_source(urandom)(3 * 8)) for _ in range(w)]
            mine = coeffs.ptr + 8 * at * stride
            raw_ntt(d_in.ptr, h, h, mine, stride, h.bit_length() - 1, w, _inv(t.omicron.value), 1, _inv(h), stream)
            if rand is not None:
                _lib.check(lib.bfs_poly_randomize(mine, stride, h, w, omega, (_u64 * w)(*[int(v) for v in rand]), stream))
        inputs.append(d_in)
        at += w
    masks = None
    if extension:      # see Table.ext_sharing_moduli: a per-polynomial summary of the support
        # (read back BEFORE the big transform is queued: the read waits for what is in front of it on the stream, and the transform --
        # most of this function's GPU time -- then runs while the caller goes on with host work)
        raw = (_u64 * total)()
        _lib.check(lib.bfs_poly_support(coeffs.ptr, stride, n_in, total, raw, stream))
        masks = [int(v) for v in raw]
    if log_n < 20:
        raw_ntt(coeffs.ptr, n_in, stride, out.ptr, n, log_n, total, omega, offset, 1, stream)
    else:
        # large domains: one call per run of tables with the same coefficient count (csrc/prover.cpp: lde_all_tables does the same on the
        # native path) -- a transform's cost depends on its zero padding (ntt_plan.hpp: 2^16 + 1 coefficients on 2^22 points take the
        # two-pass expansion plan, 2^17 + 1 the plain three passes), and a table must not pay for its neighbour's height
        k, first = 0, 0
        while k < len(tables):
            count = tables[k].height + (1 if tables[k].num_randomizers else 0) if tables[k].height else 0
            u, cols = k, 0
            while u < len(tables) and (tables[u].height + (1 if tables[u].num_randomizers else 0) if tables[u].height else 0) == count:
                cols += widths[u]
                u += 1
            if cols:
                raw_ntt(coeffs.ptr + 8 * first * stride, min(max(count, 1), n_in), stride, out.ptr + 8 * first * n, n, log_n, cols, omega, offset, 1, stream)
            first += cols
            k = u
    from .device import DeviceView
    at = 0
    for t, w, d_in in zip(tables, widths, inputs):
        view = DeviceView(out, at * n, w * n)
        t._n = n
        if extension:
            t.ext_codewords = view
            t._coefficients = masks[at:at + w] if t.height else None
        else:
            t.base_codewords = view
            t._base_device = d_in          # the trace columns stay in HBM for extend_device()
        t._last_input = None
        at += w


def zerofier_inverses(tables, domain, rows=None):
    """bfs_zerofier_inverses for a set of tables: one kernel inverts every distinct zerofier denominator of the proof at every point.
    Returns (buffer, {table: (addr of 1/(x-1), addr of 1/(x - omicron^-1), addr of 1/(x^h - 1) or None)}).
    rows = (first, count): only those points of the domain (a rank of a cooperative proof); the codewords are still n words long."""
    lib, stream = _lib.load(), current_stream()
    n = domain.length
    first, count = (0, n) if rows is None else rows
    specs = [(0, 1)]
    for t in tables:
        for spec in ((0, _inv(t.omicron.value)), (1, t.height.bit_length() - 1) if t.height else None):
            if spec is not None and spec not in specs:
                specs.append(spec)
    assert len(specs) <= 12
    out = DeviceBuffer(len(specs) * n)
    _lib.check(lib.bfs_zerofier_inverses_rows(n.bit_length() - 1, domain.offset.value, domain.omega.value, len(specs),
                                              (ctypes.c_uint32 * len(specs))(*[s[0] for s in specs]), (_u64 * len(specs))(*[s[1] for s in specs]),
                                              out.ptr, first, count, stream))
    where = {spec: out.ptr + 8 * k * n for k, spec in enumerate(specs)}
    per_table = {t: (where[(0, 1)], where[(0, _inv(t.omicron.value))],
                     where[(1, t.height.bit_length() - 1)] if t.height else None) for t in tables}
    return out, per_table


def _columns_of(values, width, out=None, out_stride=None):
    """the first `width` columns of a (rows, >= width) uint64 matrix as a (width, rows) array -- into `out` (rows out_stride apart)
    when given.  Row-major C arrays (what the native VM returns) go through bfs_host_transpose: numpy's strided copy of the same
    37 254 x 7 matrix takes three times as long, and padding is host time the GPU waits for."""
    rows = values.shape[0]
    if out is None:
        out, out_stride = np.empty((width, rows), dtype=np.uint64), rows
    if rows and width:
        if values.dtype == np.uint64 and values.flags.c_contiguous and values.ndim == 2:
            _lib.check(_lib.load().bfs_host_transpose(values.ctypes.data, rows, values.shape[1], width, out.ctypes.data, out_stride))
        else:
            out[:, :rows] = np.asarray(values[:, :width], dtype=np.uint64).T
    return out


def staging_empty(shape):
    """uint64 array for data on its way to HBM: pinned memory from the library's pool when there is a GPU, plain numpy otherwise"""
    if _POOLS.get("pinned", True):
        try:
            from .device import pinned_empty
            return pinned_empty(shape)
        except Exception:
            _POOLS["pinned"] = False
    return np.empty(shape, dtype=np.uint64)


class _PaddedMatrix:
    """the caller's matrix followed by padding rows; padding rows exist as integers and turn into element objects only if
    somebody looks at them (`matrix` stays a sequence of rows, as in the reference; the caller's row objects are kept)"""

    def __init__(self, original, array, field):
        self._original, self._array, self._field, self._made = original, array, field, {}
        self._count = len(original)

    def __len__(self):
        return self._array.shape[1]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if i < self._count:
            return self._original[i]
        if i not in self._made:
            self._made[i] = [BaseFieldElement(int(v), self._field) for v in self._array[:, i]]
        return self._made[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Table:
    air = None           # the table's constraint set (air.TableAir)
    table_index = -1     # position in BrainfuckStark.tables = index understood by bfs_air_quotients

    def __init__(self, field, base_width, full_width, length, num_randomizers, generator, order):
        assert num_randomizers in (0, 1), "one randomizer per column, as in the reference (brainfuck_stark.py:48)"
        self.field = field
        self.base_width = base_width
        self.full_width = full_width
        self.length = length
        self.num_randomizers = num_randomizers
        self.height = Table.roundup_npo2(length)
        self.omicron = Table.derive_omicron(generator, order, self.height)
        self.generator = generator
        self.order = order
        self.matrix = []
        self.ext_columns = None      # after extend(): per extension column a uint64 array (3, rows) of limb planes
        self.base_codewords = None   # DeviceBuffer, base_width * N
        self.ext_codewords = None    # DeviceBuffer, (full_width - base_width) * 3 * N
        self._coefficients = None    # host copy of the last interpolants (ext_sharing_moduli)
        self._base_device = None     # DeviceBuffer, base_width * height: the trace columns as uploaded by lde()
        self._ext_device = None      # DeviceBuffer, (full_width - base_width) * 3 * height: extension columns made by extend_device()

    @staticmethod
    def roundup_npo2(integer):
        if integer == 0 or integer == 1:
            return integer
        return 1 << (integer - 1).bit_length()

    @staticmethod
    def derive_omicron(generator, generator_order, target_order):
        while generator_order != target_order:
            generator = generator ^ 2
            generator_order = generator_order // 2
        return generator

    def unit_distance(self, omega_order):
        return 0 if self.height == 0 else omega_order // self.height

    def get_interpolation_domain_length(self):
        return self.height + self.num_randomizers

    def interpolant_degree(self):
        return self.get_interpolation_domain_length() - 1

    # ---- rows
    def base_array(self):
        """base columns as a uint64 array of shape (base_width, rows).  Converted once per matrix: from the `values` array a
        TraceMatrix of this package's VM carries, or element by element for plain lists of rows (the reference's format)."""
        if not self._array_current():
            values = getattr(self.matrix, "values", None)
            if values is not None:
                arr = _columns_of(values, self.base_width)
            elif len(self.matrix):
                arr = np.array([[_val(v) for v in row[:self.base_width]] for row in self.matrix], dtype=np.uint64).T.copy()
            else:
                arr = np.zeros((self.base_width, 0), dtype=np.uint64)
            self._set_array(arr.reshape(self.base_width, len(self.matrix)))
        return self._array

    # The column-major copy of `matrix` (and everything derived from it: scan masks, the masks prepare_extension uploaded) is cached
    # per MATRIX OBJECT.  The cache holds a reference to that object and compares with `is`, and hands out a generation number that is
    # never reused as its key: id() of a freed matrix can come back for the next one of the same height (round-3 advice).
    def _array_current(self):
        return getattr(self, "_array_for", None) is self.matrix and getattr(self, "_array_rows", -1) == len(self.matrix)

    def _set_array(self, arr):
        self._array, self._array_for, self._array_rows = arr, self.matrix, len(self.matrix)
        self._array_key = next(_ARRAY_GENERATION)

    def base_rows(self):
        return self.base_array().T.tolist()

    def _scan_masks(self):
        """the row masks of this table's scans (None = every row), in the order of _scans(): they depend on the padded base columns
        only, not on the challenges, so the prover computes and uploads them while the GPU is busy with the base columns' low-degree
        extension (prepare_extension) instead of between the first commitment and the scans"""
        key = self._array_key if self._array_current() else None
        if getattr(self, "_mask_key", None) != key or key is None:
            self._masks, self._mask_key = self._make_scan_masks(), key
        return self._masks

    def _make_scan_masks(self):
        return [None] * (self.full_width - self.base_width)

    def _rows_and_last(self):
        """(number of rows, base values of the last row or None) without materialising the column-major array when the matrix
        carries its values row-major (then _pad_to transposes straight into the padded staging array)"""
        values = getattr(self.matrix, "values", None)
        if values is not None and not self._array_current():
            rows = values.shape[0]
            return rows, ([int(v) for v in values[-1, :self.base_width]] if rows else None)
        m = self.base_array()
        return m.shape[1], ([int(v) for v in m[:, -1]] if m.shape[1] else None)

    def _pad_to(self, padding):
        """append `padding` (uint64 array, base_width x k) to the matrix"""
        values = getattr(self.matrix, "values", None)
        rows = len(self.matrix)
        fresh = not self._array_current()
        if values is not None and fresh and padding.shape[1]:
            # straight from the VM's row-major matrix into the padded staging array: one pass instead of transpose + copy
            arr = staging_empty((self.base_width, rows + padding.shape[1]))
            _columns_of(values, self.base_width, out=arr, out_stride=arr.shape[1])
        else:
            base = self.base_array()
            arr = staging_empty((base.shape[0], base.shape[1] + padding.shape[1]))      # goes to HBM as it is (Table.lde)
            arr[:, :base.shape[1]] = base
        arr[:, rows:] = padding
        self.matrix = _PaddedMatrix(self.matrix, arr, self.field)
        self._set_array(arr)

    @staticmethod
    def _counting(last, k):
        """last + 1, last + 2, ..., last + k modulo p as a uint64 array"""
        if last + k < P:
            return np.arange(last + 1, last + k + 1, dtype=np.uint64) if k else np.zeros(0, dtype=np.uint64)
        return np.array([(last + 1 + j) % P for j in range(k)], dtype=np.uint64)

    @staticmethod
    def _padding_length(rows):
        """rows to add so that the count becomes a power of two (0 and powers of two stay: `while len & (len - 1)`)"""
        return 0 if rows & (rows - 1) == 0 else (1 << rows.bit_length()) - rows

    @staticmethod
    def scan(kind, columns, mask, constants, initial, record_before):
        """bfs_xfe_scan: running product (kind 0) or running evaluation (kind 1) over the rows.  columns: up to three uint64
        arrays (None = absent); mask: bool array or None.  Returns (states as uint64 array (rows, 3), terminal triple)."""
        n = next((len(c) for c in columns if c is not None), 0 if mask is None else len(mask))
        cols = [np.ascontiguousarray(c, dtype=np.uint64) if c is not None else None for c in columns] + [None] * (3 - len(columns))
        m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
        out = np.empty((3, n), dtype=np.uint64)
        flat = [v for c in constants for v in c] + [0] * (12 - 3 * len(constants))
        terminal = (_u64 * 3)()
        _lib.check(_lib.load().bfs_xfe_scan(kind, *[c.ctypes.data if c is not None else None for c in cols[:3]],
                                            m.ctypes.data if m is not None else None, n, (_u64 * 12)(*flat), (_u64 * 3)(*initial),
                                            1 if record_before else 0, out.ctypes.data, terminal))
        return out, (int(terminal[0]), int(terminal[1]), int(terminal[2]))

    # ---- column extensions.  A table describes its running products / evaluations as scan specs; `extend` runs them with the
    # host primitive (bfs_xfe_scan), `extend_device` with the GPU one (bfs_xfe_scan_device) on the trace columns that
    # Table.lde left in HBM -- the prover's path: the extension columns then never exist on the host.
    def _scans(self, all_challenges, all_initials):
        """-> list of dicts: kind (0 product / 1 evaluation), cols (base column indices, up to three), mask (bool array or
        None), constants (up to four triples), initial (triple), before (record the state before the row's update),
        shift1 (read the first column `shift1` rows ahead, cyclically)"""
        raise NotImplementedError

    def _after_extend(self, terminals, all_challenges, read):
        """terminals: final state of every scan; read(k, row) -> the triple of extension column k at `row`"""
        raise NotImplementedError

    def extend(self, all_challenges, all_initials):
        m = self.base_array()
        futures = []
        for sp in self._scans(all_challenges, all_initials):
            cols = [m[c] for c in sp["cols"]]
            if sp.get("shift1") and len(cols[0]):
                cols[0] = np.roll(cols[0], -sp["shift1"])
            futures.append(self.scan_async(sp["kind"], cols, sp["mask"], sp["constants"], sp["initial"], sp["before"]))
        results = [f.result() for f in futures]
        self.ext_columns = [r[0] for r in results]
        self._ext_device = None
        self._after_extend([r[1] for r in results], all_challenges, lambda k, row: tuple(int(v) for v in self.ext_columns[k][:, row]))

    def extend_device(self, all_challenges, all_initials):
        """the same columns, computed in HBM from the base columns of the last lde(); needs lde() first"""
        extend_tables_device([self], all_challenges, all_initials)

    def _terminal_reads(self):
        """(extension column, row) pairs whose value _after_extend will ask for through `read` (fetched in one round trip)"""
        return []

    @staticmethod
    def scan_async(*args):
        """Table.scan on a worker thread (the native scan releases the GIL; the nine scans of a proof are independent) -> Future"""
        columns, mask = args[1], args[2]
        rows = next((len(c) for c in columns if c is not None), 0 if mask is None else len(mask))
        if rows < _THREAD_ROWS:                  # short traces: waking a worker costs more than the scan
            from concurrent.futures import Future
            done = Future()
            done.set_result(Table.scan(*args))
            return done
        return _scan_pool().submit(Table.scan, *args)

    # ---- interpolation + low-degree extension (table.py:112-148)
    def _extend_columns(self, domain, columns, randomizers, keep_coefficients=False):
        """columns: uint64 array (ncol, height); randomizers: ncol values at the point omega (or None).
        Returns the codewords over `domain` as one DeviceBuffer of ncol * N words."""
        lib, stream = _lib.load(), current_stream()
        n = domain.length
        self._n = n
        log_n = n.bit_length() - 1
        ncol, h = (columns.count // self.height if self.height else 0) if isinstance(columns, DeviceBuffer) else columns.shape[0], self.height
        out = DeviceBuffer(ncol * n)
        self._coefficients = None
        if ncol == 0:
            return out
        if h == 0:
            _lib.check(lib.bfs_memset(out.ptr, 0, out.nbytes, stream))
            return out
        omega, offset = domain.omega.value, domain.offset.value
        coeffs = DeviceBuffer(ncol * (h + 1))
        _lib.check(lib.bfs_memset(coeffs.ptr, 0, coeffs.nbytes, stream))
        d_in = columns if isinstance(columns, DeviceBuffer) else DeviceBuffer.from_numpy(columns.reshape(-1))
        self._last_input = d_in
        omicron_inv = _inv(self.omicron.value)
        raw_ntt(d_in.ptr, h, h, coeffs.ptr, h + 1, h.bit_length() - 1, ncol, omicron_inv, 1, _inv(h), stream)
        n_in = h
        self._coefficients = None
        if randomizers is not None:
            vals = (_u64 * ncol)(*[int(v) for v in randomizers])
            _lib.check(lib.bfs_poly_randomize(coeffs.ptr, h + 1, h, ncol, omega, vals, stream))
            n_in = h + 1
        assert n_in <= n, "interpolant does not fit the FRI domain"
        raw_ntt(coeffs.ptr, n_in, h + 1, out.ptr, n, log_n, ncol, omega, offset, 1, stream)
        if keep_coefficients:      # see ext_sharing_moduli: only a per-polynomial summary of the support comes back
            masks = (_u64 * ncol)()
            _lib.check(lib.bfs_poly_support(coeffs.ptr, h + 1, n_in, ncol, masks, stream))
            self._coefficients = [int(v) for v in masks]
        return out

    def ext_sharing_moduli(self, n):
        """Object identity inside the reference's extension codewords.  Its recursive ntt() adds `evens + w^i * odds`
        with Polynomial.__add__, which hands back the OTHER operand's coefficient objects when one summand is zero
        (univariate.py:23-27).  If every non-zero coefficient index of a column's interpolant is a multiple of 2^v, the
        odd halves of the top v recursion levels are all zero, so codeword elements i and i' with i = i' mod N / 2^v end
        up holding the very same BaseFieldElement objects -- visible in the proof because pickle memoises by identity
        (an all-zero trace column plus its randomizer gives c (X^h - 1): 2^v = h, N / 2^v = the table's unit distance).
        Returns, per extension column, that modulus, 1 for a constant polynomial, or None (no sharing)."""
        width = self.full_width - self.base_width
        if self._coefficients is None:
            return [None] * width
        out = []
        for c in range(width):
            mask = self._coefficients[3 * c] | self._coefficients[3 * c + 1] | self._coefficients[3 * c + 2]   # any limb non-zero
            low = mask & ((1 << 63) - 1)
            if mask == 0:
                out.append(None)                     # the zero polynomial: elements without coefficients
            elif low == 0:
                out.append(1)                        # a constant: every element holds the same coefficient objects
            else:
                v = (low & -low).bit_length() - 1
                out.append(n >> v if v else None)
        return out

    def lde(self, domain):
        cols = self.base_array().reshape(self.base_width, self.height)
        rand = None
        if self.height != 0 and self.num_randomizers:
            rand = [sample_base(random_source(urandom)(3 * 8)) for _ in range(self.base_width)]
        self._last_input = None
        self.base_codewords = self._extend_columns(domain, cols, rand)
        self._base_device = self._last_input          # the trace columns stay in HBM for extend_device()
        self._last_input = None
        return self.base_codewords

    def ldex(self, domain, xfield=None):
        width = self.full_width - self.base_width
        if self.height and getattr(self, "_ext_device", None) is not None:   # after extend_device(): already in HBM, (column, limb) planes
            cols = self._ext_device
        elif self.height:      # ext_columns: one (3, rows) array per extension column -> (column, limb) planes
            cols = staging_empty((3 * width, self.height))
            np.concatenate(self.ext_columns, axis=0, out=cols)
        else:
            cols = np.zeros((width * 3, 0), dtype=np.uint64)
        rand = None
        if self.height != 0 and self.num_randomizers:
            rand = []
            for _ in range(width):
                rand.extend(sample_ext(random_source(urandom)(3 * 8)))
        self.ext_codewords = self._extend_columns(domain, cols, rand, keep_coefficients=True)
        self._last_input = None
        return self.ext_codewords

    def ext_codeword_ptr(self, column):
        """device pointer of extension column `column` (full-width numbering): three limb planes of N words"""
        return self.ext_codewords.ptr + 8 * 3 * (column - self.base_width) * self._n

    # ---- quotients (table.py:148-281)
    def air_params(self, challenges):
        return []

    def all_quotients(self, domain, codewords, challenges, terminals):
        """codewords: ignored (the table's own codewords in HBM are used).  Returns a DeviceBuffer holding
        num_quotients extension codewords, boundary / transition / terminal order (table.py:274-281)."""
        lib, stream = _lib.load(), current_stream()
        n = domain.length
        nq = lib.bfs_air_num_quotients(self.table_index)
        out = DeviceBuffer(nq * 3 * n)
        ch = (_u64 * 33)(*[v for c in challenges for v in c])
        tm = (_u64 * 15)(*[v for t in terminals for v in t])
        params = self.air_params(challenges)
        pr = (_u64 * 3)(*params[0]) if params else None
        omicron_inv = _inv(self.omicron.value)
        _lib.check(lib.bfs_air_quotients(self.table_index, self.base_codewords.ptr, self.ext_codewords.ptr, out.ptr,
                                         n.bit_length() - 1, self.unit_distance(n), self.height, omicron_inv,
                                         domain.offset.value, domain.omega.value, ch, tm, pr, stream))
        return out

    def combine_into(self, domain, challenges, terminals, weights, accumulator, randomizer=None, randomizer_weight=None, inverses=None, rows=None):
        """bfs_air_combine: add this table's share of the non-linear combination -- its base columns, extension columns and
        quotients (in that order; weights: list of (wa, wb, shift)) -- to `accumulator` (XArray) without writing the quotient
        codewords.  randomizer (XArray) given: the accumulator is initialised to randomizer_weight * randomizer first.
        inverses: device addresses of the codewords 1/(x - 1), 1/(x - omicron^-1), 1/(x^height - 1) (zerofier_inverses), or None.
        rows = (first, count): only those points (bfs_air_combine_rows)."""
        lib, stream = _lib.load(), current_stream()
        n = domain.length
        first, count = (0, n) if rows is None else rows
        assert len(weights) == self.full_width - self.base_width + self.base_width + self.num_quotients()
        if isinstance(weights, np.ndarray):          # rows of seven words = bfs_comb_weight, as the prover lays them out
            weights = np.ascontiguousarray(weights, dtype=np.uint64)
            assert weights.shape[1] == 7
            ws = ctypes.c_void_p(weights.ctypes.data)
        else:
            ws = (_lib.CombWeight * len(weights))()
            for w, (wa, wb, shift) in zip(ws, weights):
                w.wa, w.wb, w.shift = (_u64 * 3)(*wa), (_u64 * 3)(*wb), shift
        ch = (_u64 * 33)(*[v for c in challenges for v in c])
        tm = (_u64 * 15)(*[v for t in terminals for v in t])
        params = self.air_params(challenges)
        pr = (_u64 * 3)(*params[0]) if params else None
        omicron_inv = _inv(self.omicron.value)
        _lib.check(lib.bfs_air_combine_rows(self.table_index, self.base_codewords.ptr, self.ext_codewords.ptr, n.bit_length() - 1,
                                            self.unit_distance(n), self.height, omicron_inv, domain.offset.value, domain.omega.value, ch, tm, pr,
                                            ws, randomizer.ptr if randomizer is not None else None,
                                            (_u64 * 3)(*randomizer_weight) if randomizer is not None else None, accumulator.ptr,
                                            (ctypes.c_void_p * 3)(*inverses) if inverses is not None else None, first, count, stream))

    _challenge_analysis = (None, None, None, None)
    _exact_totals = {}        # (table, kind, the values themselves) -> total degrees per constraint
    _generic_bounds = {}      # ((table, kind, zero pattern), interpolant degree) -> the bounds themselves
    _generic_totals = {}      # (table, kind, which challenges / terminals / parameters are zero) -> total degrees per constraint

    def _constraint_total_degrees(self, kind, challenges, terminals, params):
        """per constraint, the set of total degrees of the monomials that survive the exact expansion (air.expand)"""
        nvars = 2 * self.full_width if kind == "transition" else self.full_width
        memo = {}            # sub-expressions (deselectors, instruction zerofier, row differences) are shared between constraints
        return [air.total_degrees(air.expand(e, nvars, challenges, terminals, params, memo)) for e in dict(self.air.all())[kind]]

    def _degree_bounds(self, kind, challenges, terminals, exact=False):
        """exact: never take the generic shortcut (the verifier's terminals come from the prover, who chooses them AFTER it has seen the
        challenges and could make one cancel a leading monomial while still looking sampled -- round-5 advice).
        symbolic degree bounds of the constraints composed with the interpolants (multivariate.py:144-170).  Which
        monomials survive depends on the numeric challenges only through cancellations; for values that look sampled (pairwise
        distinct, zero or with more than 32 significant bits) the surviving set is the generic one except with probability
        ~2^-160, so it is computed once per zero pattern and reused.  Crafted values (the constructor's all-ones challenges,
        brainfuck_stark.py:84-92, or small test values) always take the exact expansion."""
        md = self.interpolant_degree()
        params = self.air_params(challenges)
        # (a proof asks fifteen times with the same challenges list: its part of the analysis is kept while that list is the argument)
        # -- only for a tuple of tuples, which cannot change between the calls
        seen = Table._challenge_analysis          # (one tuple, swapped whole: provers in other threads read a consistent one)
        if seen[0] is not challenges or type(challenges) is not tuple or not all(type(v) is tuple for v in challenges):
            ch_values = [tuple(v) for v in challenges]
            ch_nonzero = [v for v in ch_values if any(v)]
            seen = Table._challenge_analysis = (challenges, ch_values, set(ch_nonzero),
                                                len(set(ch_nonzero)) == len(ch_nonzero) and all(v[0] >> 32 or v[1] or v[2] for v in ch_nonzero))
        rest = [tuple(v) for v in list(terminals) + list(params)]
        rest_nonzero = [v for v in rest if any(v)]
        values = seen[1] + rest
        generic = (not exact and seen[3] and len(set(rest_nonzero)) == len(rest_nonzero) and seen[2].isdisjoint(rest_nonzero)
                   and all(v[0] >> 32 or v[1] or v[2] for v in rest_nonzero))
        if generic:
            key = (type(self).__name__, self.table_index, kind, tuple(any(v) for v in values))
            bounds = Table._generic_bounds.get((key, md))
            if bounds is not None:
                return list(bounds)
            totals = Table._generic_totals.get(key)
            if totals is None:
                totals = Table._generic_totals[key] = self._constraint_total_degrees(kind, challenges, terminals, params)
        else:
            # crafted values: the exact expansion, remembered per value tuple (the constructor asks with all-ones challenges every time
            # a prover or verifier object is made: 2 ms of symbolic expansion)
            exact_key = (type(self).__name__, self.table_index, kind, tuple(values))
            totals = Table._exact_totals.get(exact_key)
            if totals is None:
                if len(Table._exact_totals) > 256:
                    Table._exact_totals.clear()
                totals = Table._exact_totals[exact_key] = self._constraint_total_degrees(kind, challenges, terminals, params)
        bounds = [max([-1] + [t * md for t in ts]) for ts in totals]
        if generic:
            if len(Table._generic_bounds) > 4096:
                Table._generic_bounds.clear()
            Table._generic_bounds[(key, md)] = tuple(bounds)
        return bounds

    def boundary_quotient_degree_bounds(self, challenges):
        return [b - 1 for b in self._degree_bounds("boundary", challenges, [air.X0] * 5)]

    def transition_quotient_degree_bounds(self, challenges):
        return [b - self.height + 1 for b in self._degree_bounds("transition", challenges, [air.X0] * 5)]

    def terminal_quotient_degree_bounds(self, challenges, terminals, exact=False):
        return [b - 1 for b in self._degree_bounds("terminal", challenges, terminals, exact)]

    def all_quotient_degree_bounds(self, challenges, terminals, exact_terminals=False):
        return (self.boundary_quotient_degree_bounds(challenges) + self.transition_quotient_degree_bounds(challenges)
                + self.terminal_quotient_degree_bounds(challenges, terminals, exact_terminals))

    def num_quotients(self, challenges=None, terminals=None):
        return sum(len(c) for _, c in self.air.all())

    def max_transition_degree(self, challenges):
        """symbolic degree of the composed transition constraints minus the zerofier (brainfuck_stark.py:84-92)"""
        return max([b - (self.height - 1) for b in self._degree_bounds("transition", challenges, [air.X0] * 5)] or [1])

    # ---- the constraints as the reference presents them: lists of MPolynomial (table.py:145-146, 173-174, 248-249)
    def _constraints_ext(self, kind, challenges, terminals=None):
        from .extension_field import ExtensionField
        from .multivariate import MPolynomial
        xfield = challenges[0].field if hasattr(challenges[0], "field") else ExtensionField.main()

        def triple(v):
            return tuple(v.limbs()) if hasattr(v, "limbs") else tuple(v)
        ch = [triple(c) for c in challenges]
        tm = [triple(t) for t in terminals] if terminals is not None else [air.X0] * 5
        nvars = 2 * self.full_width if kind == "transition" else self.full_width
        memo, out = {}, []
        for e in dict(self.air.all())[kind]:
            expansion = air.expand(e, nvars, ch, tm, self.air_params(ch), memo)
            out.append(MPolynomial({air.unpack_exponents(k, nvars): xfield.from_limbs(list(v)) for k, v in expansion.items()}))
        return out

    def boundary_constraints_ext(self, challenges):
        return self._constraints_ext("boundary", challenges)

    def transition_constraints_ext(self, challenges):
        return self._constraints_ext("transition", challenges)

    def terminal_constraints_ext(self, challenges, terminals):
        return self._constraints_ext("terminal", challenges, terminals)

    # ---- host-side evaluation at one point (the verifier's use, table.py:283-311)
    def evaluate_all_constraints(self, point, next_point, challenges, terminals):
        """(boundary values, transition values, terminal values) at one point through the generated constraint code compiled for
        the host (bfs_air_evaluate) -- the same straight-line code the kernels run; evaluate_constraints below walks the expression
        graphs in Python and is what the tests compare this with.  point / next_point: base columns then extension columns, as
        limb triples."""
        lib = _lib.load()
        bw, xw = self.base_width, self.full_width - self.base_width
        nb, nt, nz = (len(c) for _, c in self.air.all())
        params = self.air_params(challenges)
        out = (_u64 * (3 * (nb + nt + nz)))()
        _lib.check(lib.bfs_air_evaluate(
            self.table_index, (_u64 * bw)(*[p[0] for p in point[:bw]]), (_u64 * bw)(*[p[0] for p in next_point[:bw]]),
            (_u64 * (3 * xw))(*[v for p in point[bw:] for v in p]), (_u64 * (3 * xw))(*[v for p in next_point[bw:] for v in p]),
            (_u64 * 33)(*[v for c in challenges for v in c]), (_u64 * 15)(*[v for t in terminals for v in t]),
            (_u64 * 3)(*params[0]) if params else None, out))
        vals = [tuple(out[3 * q:3 * q + 3]) for q in range(nb + nt + nz)]
        return vals[:nb], vals[nb:nb + nt], vals[nb + nt:]

    def evaluate_constraints(self, kind, point, next_point, challenges, terminals):
        cons = dict(self.air.all())[kind]
        params = self.air_params(challenges)
        return [air.evaluate(e, point, next_point, challenges, terminals, params) for e in cons]
