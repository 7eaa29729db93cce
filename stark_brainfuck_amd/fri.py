"""FRI low-degree test, prover on the GPU -- mirror of the reference's `fri.py` (/root/reference/code/fri.py:13-319).

    Fri(offset, omega, initial_domain_length, expansion_factor, num_colinearity_tests, xfield)
      .domain   Fri.Domain: offset, omega, length, __call__, list, evaluate, xevaluate, interpolate, xinterpolate
      .num_rounds()  .sample_indices(...)  .commit(...)  .query(...)  .query_last(...)  .prove(...)  .verify(...)

`prove` / `commit` run the whole round loop natively (csrc/fri.hip): Merkle trees, folding and openings on the GPU,
Fiat-Shamir on the host in C++.  `codeword` may be a Python list of ExtensionFieldElement (as in the reference) or an
XArray already in HBM.  `verify` is the verifier: host-side, as in the reference.
"""
import ctypes
from hashlib import blake2b

from . import _lib
from .arrays import BaseArray, XArray
from .device import current_stream
from .ip import NativeTranscript, ProofStream
from .merkle import Merkle
from .ntt import _base_value, _transform, fast_coset_interpolate
from .univariate import Polynomial, colinear

_u64 = ctypes.c_uint64


class _Codeword:
    """a round codeword living in HBM that behaves like the reference's list of elements (lazy, identity-stable)."""

    def __init__(self, xarray):
        self.array = xarray
        self._items = None

    def __len__(self):
        return self.array.n

    def _all(self):
        if self._items is None:
            self._items = self.array.to_elements()
        return self._items

    def __getitem__(self, i):
        return self._all()[i]

    def __iter__(self):
        return iter(self._all())


class Fri:
    class Domain:
        def __init__(self, offset, omega, length):
            self.offset = offset
            self.omega = omega
            self.length = length

        def __call__(self, index):
            return (self.omega ^ index) * self.offset

        def list(self):
            out, x = [], self.offset
            for _ in range(self.length):
                out.append(x)
                x = x * self.omega
            return out

        def _evaluate(self, polynomial, as_array):
            coeffs = polynomial.coefficients if isinstance(polynomial, Polynomial) else polynomial
            if isinstance(coeffs, (XArray, BaseArray)):
                src, n_in = coeffs, len(coeffs)
            else:
                assert len(coeffs) <= self.length, "polynomial has more coefficients than the domain has points"
                if not coeffs:
                    return [self.omega.field.zero() for _ in range(self.length)]
                from .extension_field import ExtensionFieldElement
                src = XArray.from_elements(coeffs) if isinstance(coeffs[0], ExtensionFieldElement) else BaseArray.from_elements(coeffs)
                n_in = len(coeffs)
            out = _transform(src, n_in, self.length, _base_value(self.omega), _base_value(self.offset), 1)
            return out if as_array else out.to_elements()

        def evaluate(self, polynomial, as_array=False):
            """coset evaluation of a base-field polynomial (fri.py:26-30)."""
            return self._evaluate(polynomial, as_array)

        def xevaluate(self, polynomial, xfield=None, as_array=False):
            """coset evaluation of an extension-field polynomial (fri.py:32-37): three limb transforms."""
            if xfield is None and isinstance(polynomial, Polynomial):
                assert len(polynomial.coefficients) != 0, "trying to xevaluate zero polynomial with no target field"
            return self._evaluate(polynomial, as_array)

        def interpolate(self, values):
            return fast_coset_interpolate(self.offset, self.omega, values)

        def xinterpolate(self, values):
            return fast_coset_interpolate(self.offset, self.omega, values)

    def __init__(self, offset, omega, initial_domain_length, expansion_factor, num_colinearity_tests, xfield):
        self.domain = Fri.Domain(offset, omega, initial_domain_length)
        self.field = xfield
        self.expansion_factor = expansion_factor
        self.num_colinearity_tests = num_colinearity_tests
        assert self.num_rounds() >= 1, "cannot do FRI with less than one round"

    def num_rounds(self):
        length, rounds = self.domain.length, 0
        while length > self.expansion_factor:
            length //= 2
            rounds += 1
        return rounds

    @staticmethod
    def sample_index(byte_array, size):
        return int.from_bytes(bytes(byte_array), "big") % size

    def sample_indices(self, seed, size, reduced_size, number):
        assert number <= reduced_size, \
            f"cannot sample more indices than available in last codeword; requested: {number}, available: {reduced_size}"
        assert number <= 2 * reduced_size, "not enough entropy in indices wrt last codeword"
        indices, reduced, counter = [], set(), 0
        while len(indices) < number:
            index = Fri.sample_index(blake2b(seed + bytes(counter)).digest(), size)
            counter += 1
            if index % reduced_size not in reduced:
                indices.append(index)
                reduced.add(index % reduced_size)
        return indices

    def eval_domain(self):
        return self.domain.list()

    # ------------------------------------------------------------------ prover (GPU)
    def _as_xarray(self, codeword):
        if isinstance(codeword, _Codeword):
            return codeword.array
        return codeword if isinstance(codeword, XArray) else XArray.from_elements(list(codeword))

    def _run_native(self, codeword, proof_stream, with_query, known_leafs=None, round0_tree=None):
        lib, stream = _lib.load(), current_stream()
        arr = self._as_xarray(codeword)
        n = len(arr)
        assert n & (n - 1) == 0, "codeword length must be a power of two"
        transcript = proof_stream._native() if hasattr(proof_stream, "_native") else None
        if transcript is None or (transcript.xfield is not None and transcript.xfield is not self.field):
            transcript = NativeTranscript()
            transcript.xfield = self.field
            transcript.scan(proof_stream.objects)
            for o in proof_stream.objects:
                transcript.push(o)
        elif transcript.xfield is None:
            transcript.xfield = self.field
        before = transcript.num_objects()
        session = lib.bfs_fri_session_new()
        try:
            if round0_tree is not None and round0_tree._nodes_host is None and round0_tree.num_leafs == n:
                _lib.check(lib.bfs_fri_session_round0_tree(session, round0_tree._nodes.ptr, round0_tree.root()))
            _lib.check(lib.bfs_fri_commit(session, transcript.handle, arr.ptr, arr.stride, n.bit_length() - 1,
                                          _base_value(self.domain.offset), _base_value(self.domain.omega), self.expansion_factor, stream))
            top = None
            for index, obj in (known_leafs or {}).items():
                # element objects of this codeword that are already in the stream keep their identity (pickle memoises by id)
                # (an int is the handle of an element native code has already put into the transcript)
                _lib.check(lib.bfs_fri_session_alias(session, transcript.handle, 0, index, obj if isinstance(obj, int) else transcript.to_native(obj)))
            if with_query:
                out = (_u64 * self.num_colinearity_tests)()
                _lib.check(lib.bfs_fri_query(session, transcript.handle, self.num_colinearity_tests, out, stream))
                top = [int(x) for x in out]
            if hasattr(proof_stream, "_adopt_lazy"):
                proof_stream._adopt_lazy(transcript, before, transcript.num_objects(), self.field)
            else:
                for i in range(before, transcript.num_objects()):
                    proof_stream.push(transcript.to_python(lib.bfs_ps_object_at(transcript.handle, i), self.field))
            rounds = []
            for r in range(lib.bfs_fri_session_rounds(session)):
                cw, nodes = ctypes.c_void_p(), ctypes.c_void_p()
                length, stride = _u64(), _u64()
                root = ctypes.create_string_buffer(64)
                _lib.check(lib.bfs_fri_session_round(session, r, ctypes.byref(cw), ctypes.byref(length), ctypes.byref(stride), ctypes.byref(nodes), root))
                rounds.append((cw.value, length.value, stride.value, nodes.value, root.raw))
            return top, rounds, session, arr
        except Exception:
            lib.bfs_fri_session_free(session)
            if getattr(proof_stream, "_cached", None) is transcript:
                proof_stream._cached = None          # native code may have appended objects the Python list does not have
            raise

    def commit(self, codeword, proof_stream, round_index=0):
        """fri.py:91-139 -> (codewords, trees); both stay in HBM and are materialised lazily."""
        lib = _lib.load()
        _, rounds, session, arr = self._run_native(codeword, proof_stream, with_query=False)
        keeper = _SessionKeeper(lib, session, arr)
        codewords, trees = [], []
        for r, (cw, length, stride, nodes, _root) in enumerate(rounds):
            view = _Codeword(XArray(_Borrowed(cw, keeper), length, self.field, stride))
            if r == 0 and isinstance(codeword, list):
                view._items = codeword
            if r + 1 == len(rounds):
                view._items = proof_stream.objects[-1]      # fri.py:134 pushes this very list: keep object identity
            codewords.append(view)
            if r + 1 < len(rounds):
                trees.append(Merkle(view, _device_nodes=_Borrowed(nodes, keeper)))
        return codewords, trees

    def query(self, current_tree, next_tree, c_indices, proof_stream):
        """fri.py:141-158"""
        half = len(current_tree.leafs) // 2
        a_indices, b_indices = list(c_indices), [i + half for i in c_indices]
        for s in range(self.num_colinearity_tests):
            proof_stream.push((current_tree.leafs[a_indices[s]], current_tree.leafs[b_indices[s]], next_tree.leafs[c_indices[s]]))
        for s in range(self.num_colinearity_tests):
            proof_stream.push(current_tree.open(a_indices[s]))
            proof_stream.push(current_tree.open(b_indices[s]))
            proof_stream.push(next_tree.open(c_indices[s]))
        return a_indices + b_indices

    def query_last(self, current_tree, last_codeword, c_indices, proof_stream):
        """fri.py:160-176"""
        half = len(current_tree.leafs) // 2
        a_indices, b_indices = list(c_indices), [i + half for i in c_indices]
        for s in range(self.num_colinearity_tests):
            proof_stream.push((current_tree.leafs[a_indices[s]], current_tree.leafs[b_indices[s]], last_codeword[c_indices[s]]))
        for s in range(self.num_colinearity_tests):
            proof_stream.push(current_tree.open(a_indices[s]))
            proof_stream.push(current_tree.open(b_indices[s]))
        return a_indices + b_indices

    def prove(self, codeword, proof_stream, known_leafs=None, round0_tree=None):
        """fri.py:178-199: commit + query in one native call; returns the top-level indices.
        known_leafs: {index: element object} for elements of `codeword` that the caller has already pushed.
        round0_tree: a Merkle tree the caller has already built over `codeword` (round 0 would build the same one)."""
        assert self.domain.length == len(codeword), "initial codeword length does not match length of initial codeword"
        top, _, session, _ = self._run_native(codeword, proof_stream, with_query=True, known_leafs=known_leafs, round0_tree=round0_tree)
        _lib.load().bfs_fri_session_free(session)
        return top

    # ------------------------------------------------------------------ verifier (host, fri.py:201-319)
    def verify(self, proof_stream, root):
        """fri.py:232-319.  Same checks in the same order; the abscissae offset * omega^i and the two polynomial questions -- "has the
        interpolant of the last codeword degree <= d" and "are these three points on a line" -- are answered on integer residues
        (an inverse transform's non-zero pattern, two cross products) instead of through Polynomial objects over element objects:
        the reference's interpolations were 2/3 of this package's 15 ms verifier."""
        from .air import P
        from .merkle import leaf_pickle_source
        if leaf_pickle_source.get() is None and hasattr(proof_stream, "pickle_of"):
            token = leaf_pickle_source.set(proof_stream.pickle_of)      # (a bare Fri.verify on a deserialised stream; see BrainfuckStark.verify)
            try:
                return self.verify(proof_stream, root)
            finally:
                leaf_pickle_source.reset(token)
        omega_v, offset_v = _base_value(self.domain.omega), _base_value(self.domain.offset)
        rounds, t, N = self.num_rounds(), self.num_colinearity_tests, self.domain.length
        roots, alphas = [root], []
        for r in range(rounds):
            if r > 0:
                roots.append(proof_stream.pull())
            alphas.append(self.field.sample(proof_stream.verifier_fiat_shamir()))
        last_codeword = proof_stream.pull()
        if roots[-1] != _host_merkle_root(last_codeword):
            print("last codeword is not well formed")
            return False
        n_last = len(last_codeword)
        degree = (n_last // self.expansion_factor) - 1
        last_omega_v = pow(omega_v, 1 << (rounds - 1), P)
        assert pow(last_omega_v, n_last, P) == 1, "omega does not have right order"
        top = _interpolant_degree(last_omega_v, [tuple(e.limbs()) for e in last_codeword])
        if top > degree:
            return False
        top_level_indices = self.sample_indices(proof_stream.verifier_fiat_shamir(), N >> 1, N >> (rounds - 1), t)
        for r in range(rounds - 1):
            half = N >> (r + 1)
            c_indices = [i % half for i in top_level_indices]
            a_indices, b_indices = list(c_indices), [i + half for i in c_indices]
            alpha = tuple(alphas[r].limbs())
            aa, bb, cc = [], [], []
            for s in range(t):
                ay, by, cy = proof_stream.pull()
                aa.append(ay); bb.append(by); cc.append(cy)
                ax, bx = offset_v * pow(omega_v, a_indices[s], P) % P, offset_v * pow(omega_v, b_indices[s], P) % P
                ya, yb, yc = tuple(ay.limbs()), tuple(by.limbs()), tuple(cy.limbs())
                on_a_line = _on_a_line(ax, ya, bx, yb, alpha, yc)
                if on_a_line is None:          # coinciding abscissae: let the general routine decide (and fail the way the reference's does)
                    from .algebra import BaseField, BaseFieldElement
                    lift, base = self.field.lift, BaseField.main()
                    on_a_line = colinear([(lift(BaseFieldElement(ax, base)), ay), (lift(BaseFieldElement(bx, base)), by), (alphas[r], cy)])
                if not on_a_line:
                    print("colinearity check failure")
                    return False
            for i in range(t):
                if not Merkle.verify(roots[r], a_indices[i], proof_stream.pull(), aa[i]):
                    print("merkle authentication path verification fails for aa")
                    return False
                if not Merkle.verify(roots[r], b_indices[i], proof_stream.pull(), bb[i]):
                    print("merkle authentication path verification fails for bb")
                    return False
                if r + 1 != rounds - 1:
                    if not Merkle.verify(roots[r + 1], c_indices[i], proof_stream.pull(), cc[i]):
                        print("merkle authentication path verification fails for cc")
                        return False
            if r + 1 == rounds - 1:
                for i in range(t):
                    if cc[i] != last_codeword[c_indices[i]]:
                        print("leafs in last round do not correspond to last codeword")
                        return False
            omega_v, offset_v = omega_v * omega_v % P, offset_v * offset_v % P
        return True


def _host_merkle_root(leaves):
    """Merkle(leaves).root() (merkle.py:8-44) with hashlib, for the verifier's check of the last codeword -- a few dozen elements.  The
    verifier runs on the host throughout, like the reference's (fri.py:201-319), so verify() needs no GPU; the prover's trees are
    built by the kernels of csrc/merkle.hip and compared with this construction by the tests."""
    from hashlib import blake2b
    from .merkle import leaf_bytes
    n = len(leaves)
    if n == 0 or n & (n - 1):
        return Merkle(leaves).root()            # (never the case for a FRI codeword; the reference's padding rules live in Merkle)
    level = [blake2b(leaf_bytes(e)).digest() for e in leaves]
    while len(level) > 1:
        level = [blake2b(level[2 * i] + level[2 * i + 1]).digest() for i in range(len(level) // 2)]
    return level[0]


def _interpolant_degree(omega, values):
    """degree of the polynomial that takes the extension-field `values` (integer triples) on the coset offset * omega^i, i < n (-1:
    the zero polynomial) -- what `Polynomial.interpolate_domain(...).degree()` answers in fri.py:253-259.  Coefficient j of the
    interpolant is, up to the non-zero factor n * offset^j, sum_i y_i omega^(-i j): the offset does not matter."""
    from .air import P
    n = len(values)
    inverse = pow(omega, P - 2, P)
    for j in range(n - 1, -1, -1):
        step, w, acc = pow(inverse, j, P), 1, (0, 0, 0)
        for y in values:
            acc = ((acc[0] + y[0] * w) % P, (acc[1] + y[1] * w) % P, (acc[2] + y[2] * w) % P)
            w = w * step % P
        if any(acc):
            return j
    return -1


def _on_a_line(ax, ya, bx, yb, cx, yc):
    """univariate.colinear for the verifier's three points (ax, ya), (bx, yb), (cx, yc) -- ax, bx base-field residues, cx and the
    ordinates extension triples: True iff the interpolant has degree exactly 1 (a non-zero slope and equal cross products); None
    when two abscissae coincide (the general routine then decides, and fails, as the reference's does)."""
    from .air import P, xmul, xscale, xsub
    d1, d2, dx = (bx - ax) % P, xsub(cx, (ax, 0, 0)), xsub(cx, (bx, 0, 0))
    if not (d1 and any(d2) and any(dx)):
        return None
    e1 = xsub(yb, ya)
    return bool(any(e1)) and xscale(xsub(yc, ya), d1) == xmul(e1, d2)


class _SessionKeeper:
    """keeps a native FRI session (and the round-0 codeword) alive while views into its HBM block exist."""

    def __init__(self, lib, session, arr):
        self.lib, self.session, self.arr = lib, session, arr

    def __del__(self):
        try:
            self.lib.bfs_fri_session_free(self.session)
        except Exception:
            pass


class _Borrowed:
    """a device pointer owned by someone else, shaped like DeviceBuffer for reads."""

    def __init__(self, ptr, keeper):
        self.ptr, self._keeper = ptr, keeper

    def to_numpy(self, count, offset=0, stream=None):
        import numpy as np
        out = np.empty(count, dtype=np.uint64)
        if count:
            _lib.check(_lib.load().bfs_memcpy_d2h(out.ctypes.data, self.ptr + 8 * offset, count * 8,
                                                  stream if stream is not None else current_stream()))
        return out
