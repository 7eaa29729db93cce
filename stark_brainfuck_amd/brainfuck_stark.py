"""STARK prover for the Brainfuck VM -- mirror of the reference's `brainfuck_stark.py`
(/root/reference/code/brainfuck_stark.py:20-341): same constructor, `prove(...)`, `get_terminals`, `sample_weights`,
`sample_indices`, and the same transcript, byte for byte.

Where the time goes in the reference, and where it goes here:
  trace interpolation + low-degree extension (:165-172, 194-195)    batched INTT / randomizer fix / coset NTT in HBM (table.py)
  table extension: running products / evaluations (:186-187)        prefix scans on the trace columns in HBM (bfs_xfe_scan_device)
  commitments to zipped codewords (:178-179, 197-198)               row pickles synthesised and hashed on the GPU (bfs_merkle_build_rows)
  quotient codewords (:204-221, 93 % of the reference's time)       one kernel per table, folded straight into ...
  non-linear combination of 151 terms (:236-298)                    ... the combination accumulator (bfs_air_combine); with
                                                                    keep_intermediates the quotients are written out (bfs_air_quotients,
                                                                    bfs_combination) so that tests can compare each with the reference's
  FRI (:336)                                                        fri.Fri.prove (bfs_fri_commit / bfs_fri_query), round 0 on the
                                                                    combination tree that was just built
Padding, Fiat-Shamir sampling, transcript assembly and the object-identity bookkeeping of opened rows stay on the host.
"""
import ctypes

import numpy as np
from hashlib import blake2b
from os import urandom          # module-level on purpose: tests patch `brainfuck_stark.urandom` for determinism
from .randomness import source as random_source

from . import _lib, air
from .algebra import BaseField, BaseFieldElement
from .arrays import XArray
from .device import GatherBatch, current_stream, gather, synchronize
from .evaluation_argument import EvaluationArgument, ProgramEvaluationArgument
from .extension_field import ExtensionField, ExtensionFieldElement
from .fri import Fri
from .instruction_table import InstructionTable
from .io_table import InputTable, OutputTable
from .ip import ProofStream
from .memory_table import MemoryTable
from .merkle import Merkle
from .permutation_argument import PermutationArgument
from .processor_table import ProcessorTable
from .salted_merkle import SaltedMerkle, ZippedSaltedMerkle
from .algebra import P_GOLDILOCKS
from .table import extend_tables_device, lde_tables, prepare_extension, sample_ext, sample_ext_many, zerofier_inverses
from .univariate import Polynomial
from .vm import VirtualMachine

_u64 = ctypes.c_uint64


class _MalformedProof(Exception):
    """an object in the proof is not what its position requires (verify() answers False)"""


class BrainfuckStark:
    field = BaseField.main()
    xfield = ExtensionField.main()

    def __init__(self, running_time, memory_length, program, input_symbols, output_symbols, log_expansion_factor=2, security_level=2):
        """the reference fixes log_expansion_factor = 2 and security_level = 2 "for speed" (brainfuck_stark.py:31-36, with 4 and 160
        commented as the real values); they are parameters here, the defaults reproduce the reference"""
        self.running_time = running_time
        self.memory_length = memory_length
        self.program = program
        self.input_symbols = input_symbols
        self.output_symbols = output_symbols

        self.expansion_factor = 1 << log_expansion_factor
        self.security_level = security_level
        self.num_colinearity_checks = self.security_level // log_expansion_factor
        assert self.expansion_factor & (self.expansion_factor - 1) == 0, "expansion factor must be a power of 2"
        assert self.expansion_factor >= 4, "expansion factor must be 4 or greater"
        assert self.num_colinearity_checks * log_expansion_factor >= self.security_level, \
            "number of colinearity checks times log of expansion factor must be at least security level"
        self.num_randomizers = 1

        order = 1 << 32
        smooth_generator = BrainfuckStark.field.primitive_nth_root(order)
        f = self.field
        self.processor_table = ProcessorTable(f, running_time, self.num_randomizers, smooth_generator, order)
        self.instruction_table = InstructionTable(f, running_time + len(program), self.num_randomizers, smooth_generator, order)
        self.memory_table = MemoryTable(f, memory_length, self.num_randomizers, smooth_generator, order)
        self.input_table = InputTable(f, len(input_symbols), smooth_generator, order)
        self.output_table = OutputTable(f, len(output_symbols), smooth_generator, order)
        self.tables = [self.processor_table, self.instruction_table, self.memory_table, self.input_table, self.output_table]

        self.permutation_arguments = [
            PermutationArgument(self.tables, (0, ProcessorTable.instruction_permutation), (1, InstructionTable.permutation)),
            PermutationArgument(self.tables, (0, ProcessorTable.memory_permutation), (2, MemoryTable.permutation))]
        self.evaluation_arguments = [
            EvaluationArgument(8, 2, [BaseFieldElement(ord(i), f) for i in input_symbols]),
            EvaluationArgument(9, 3, [BaseFieldElement(ord(o), f) for o in output_symbols]),
            ProgramEvaluationArgument([0, 1, 2, 10], 4, program)]

        # FRI domain length from the degree of the composed transition constraints (:82-95)
        self.max_degree = 1
        ones = [air.X1] * 11
        for table in self.tables:
            self.max_degree = max(self.max_degree, table.max_transition_degree(ones))
        self.max_degree = BrainfuckStark.roundup_npo2(self.max_degree) - 1
        fri_domain_length = (self.max_degree + 1) * self.expansion_factor
        generator = BrainfuckStark.field.generator()
        omega = BrainfuckStark.field.primitive_nth_root(fri_domain_length)
        self.fri = Fri(generator, omega, fri_domain_length, self.expansion_factor, self.num_colinearity_checks, self.xfield)

    def get_terminals(self):
        """the five terminals as int triples (:103-109)"""
        return [self.processor_table.instruction_permutation_terminal, self.processor_table.memory_permutation_terminal,
                self.processor_table.input_evaluation_terminal, self.processor_table.output_evaluation_terminal,
                self.instruction_table.evaluation_terminal]

    @staticmethod
    def _sample_weights(number, randomness, as_array=False):
        """:104-112: weight i = ExtensionField.sample(blake2b(randomness + bytes(i)).digest()), i.e. the three 21-byte big-endian
        chunks of the digest mod p (extension_field.py:100-111; the 64th byte is not used).  One integer conversion per digest."""
        if number > 4:             # natively (bfs_sample_weights): 157 digests and 471 reductions are ~150 us of a 4 ms proof in Python
            import ctypes
            lib = _lib.load()
            raw = (ctypes.c_uint64 * (3 * number))()
            randomness = bytes(randomness)
            _lib.check(lib.bfs_sample_weights(randomness, len(randomness), number, raw))
            if as_array:           # (number, 3) uint64: the prover hands the weights on to the kernels as they are
                return np.frombuffer(raw, dtype=np.uint64).reshape(number, 3)
            flat = list(raw)
            return [(flat[3 * i], flat[3 * i + 1], flat[3 * i + 2]) for i in range(number)]
        out, mask = [], (1 << 168) - 1
        for i in range(number):
            v = int.from_bytes(blake2b(randomness + bytes(i)).digest()[:63], "big")
            out.append(((v >> 336) % P_GOLDILOCKS, ((v >> 168) & mask) % P_GOLDILOCKS, (v & mask) % P_GOLDILOCKS))
        return np.array(out, dtype=np.uint64).reshape(number, 3) if as_array else out

    def sample_weights(self, number, randomness):
        """:111-112, as extension-field element objects"""
        return [self.xfield.from_limbs(w) for w in BrainfuckStark._sample_weights(number, randomness)]

    @staticmethod
    def sample_indices(number, randomness, bound):
        """:114-123"""
        return [int.from_bytes(blake2b(randomness + bytes(i)).digest(), "big") % bound for i in range(number)]

    @staticmethod
    def roundup_npo2(integer):
        if integer == 0 or integer == 1:
            return 1
        return 1 << (integer - 1).bit_length()

    # ------------------------------------------------------------------------------------------------------------
    keep_intermediates = False      # True: prove() leaves trees, quotient codewords and the combination codeword in `_last` (tests)
    stage_timing = False            # True: prove() synchronises its stream after every stage so that `timing` splits the GPU time by stage
                                    # (bench.py's breakdown, tools/); False: `timing` holds host time per stage and the stages overlap freely

    # ---- several GPUs on one proof (shard.RowShardedSaltedMerkle): every rank runs the polynomial stages on all columns and hashes
    # only its range of the zipped rows; set by cooperate() and used inside shard.shared_randomness()
    _cooperation = None
    _row_windows = None      # tests: [(first, count), ...] tiling the FRI domain -- the combination stage runs window by window
    _shift_tweak = None      # tests: shifts -> shifts, applied to the combination's degree shifts (both combination paths see the result)

    def cooperate(self, world_size, rank, group=None, device=None):
        """this prover is one of `world_size` identical provers (one per GPU) working on the SAME proof: the zipped commitments are
        built cooperatively, every rank returns the same proof bytes.  Call prove() inside `with shard.shared_randomness(...)`."""
        self._cooperation = (int(world_size), int(rank), group, device) if world_size > 1 else None
        return self

    def _zipped_tree(self, columns, n, make_row):
        if self._cooperation is None:
            return ZippedSaltedMerkle(columns, n, make_row)
        from .shard import RowShardedSaltedMerkle
        world_size, rank, group, device = self._cooperation
        return RowShardedSaltedMerkle(columns, n, make_row, world_size, rank, group=group, device=device)

    def _openings_python(self, proof_stream, base_tree, extension_tree, combination, combination_tree, base_requests, ext_requests,
                         fetched_base, fetched_ext, indices, unit_distances, n, xf):
        """the openings (:315-333) through Python objects: used when the transcript cannot take them natively (a foreign proof
        stream class, the row-sharded trees of a cooperative proof)"""
        # everything the openings read from HBM -- rows, salts, authentication paths, combination leaves -- in one round trip
        rows = list(dict.fromkeys((index + distance) % n for index in indices for distance in [0] + unit_distances))
        batch = GatherBatch()
        row_tickets = {i: ([batch.add(*r) for r in base_requests(i)], [batch.add(*r) for r in ext_requests(i)]) for i in rows}
        leaf_tickets = {index: batch.add(combination.ptr + 8 * index, 3, combination.stride) for index in dict.fromkeys(indices)}
        stores = [base_tree.prefetch_salts(rows, batch), extension_tree.prefetch_salts(rows, batch),
                  base_tree.prefetch_paths(rows, batch), extension_tree.prefetch_paths(rows, batch),
                  combination_tree.prefetch_paths(indices, batch)]
        batch.run()
        for store in stores:
            store()
        for i, (tb, te) in row_tickets.items():
            fetched_base[i] = np.concatenate([batch.words(t) for t in tb])
            fetched_ext[i] = np.concatenate([batch.words(t) for t in te])
        for index in indices:
            for distance in [0] + unit_distances:
                idx = (index + distance) % n
                proof_stream.push(base_tree.leafs[idx][0])
                proof_stream.push(base_tree.open(idx))
                proof_stream.push(extension_tree.leafs[idx][0])
                proof_stream.push(extension_tree.open(idx))
        known = {}
        for index in indices:
            if index not in known:                   # the same index twice is the same leaf object twice
                known[index] = xf.from_limbs([int(v) for v in batch.words(leaf_tickets[index])])
            leaf = known[index]
            proof_stream.push(leaf)
            proof_stream.push(combination_tree.open(index))

        return known

    def _openings_native(self, proof_stream, base_tree, extension_tree, combination, combination_tree, base_row, ext_row, moduli,
                         indices, unit_distances, n, f2, xf, lib, stream):
        """the same through bfs_stark_push_openings: rows, salts, paths and leaves go from HBM into the native transcript in one
        call, and become Python objects only if somebody looks at proof_stream.objects (ip.ProofStream._adopt_lazy).  Returns
        {index: handle of the combination leaf} for Fri.prove, or None when this route is not available."""
        if self._cooperation is not None or type(base_tree) is not ZippedSaltedMerkle or type(extension_tree) is not ZippedSaltedMerkle:
            return None
        if not hasattr(proof_stream, "_adopt_lazy") or combination_tree._nodes_host is not None or combination_tree.num_leafs != n:
            return None
        transcript = proof_stream._native()
        if getattr(proof_stream, "_cached", None) is not transcript or (transcript.xfield is not None and transcript.xfield is not xf):
            return None
        if transcript.xfield is None:
            transcript.xfield = xf

        def requests(reqs):
            arr = (_lib.GatherRequest * len(reqs))()
            for a, (ptr, nwords, stride) in zip(arr, reqs):
                a.d_base, a.nwords, a.stride, a.out_offset = ptr, nwords, stride, 0
            return arr

        def salts(tree):
            if getattr(tree, "_salt_cache", None) is not None:
                return ctypes.c_void_p(tree._salt_base), 1
            return ctypes.c_void_p(ctypes.addressof(tree._salt_host)), 0
        base_arr, ext_arr = requests(base_row), requests(ext_row)
        mods = (_u64 * max(len(moduli), 1))(*[0 if m is None else int(m) for m in moduli])
        idx = (_u64 * len(indices))(*indices)
        dist = (_u64 * (1 + len(unit_distances)))(0, *unit_distances)
        out = (_u64 * len(indices))()
        (bs, bs_dev), (es, es_dev) = salts(base_tree), salts(extension_tree)
        before = transcript.num_objects()
        _lib.check(lib.bfs_stark_push_openings(transcript.handle, base_arr, len(base_row), transcript._field_id(f2), ext_arr, len(ext_row), mods,
                                               len(moduli), n, base_tree._nodes.ptr, bs, bs_dev, extension_tree._nodes.ptr, es, es_dev,
                                               combination.ptr, combination.stride, combination_tree._nodes.ptr, idx, len(indices), dist, len(dist),
                                               out, stream))
        proof_stream._adopt_lazy(transcript, before, transcript.num_objects(), xf)
        return {index: int(handle) for index, handle in zip(indices, out)}

    @staticmethod
    def _release(*holders):
        """hand device memory back to the pool now instead of when the garbage collector gets to it (the blocks are
        stream-ordered: kernels already queued keep reading them, the next proof on this stream reuses them)"""
        for h in holders:
            for name in ("buf", "_nodes", "_salts", "base_codewords", "ext_codewords", "_base_device", "_ext_device"):
                b = getattr(h, name, None)
                if hasattr(b, "free"):
                    b.free()
                    if name.endswith("codewords") or name.endswith("_device"):
                        setattr(h, name, None)
            if hasattr(h, "free"):
                h.free()

    def prove(self, program, processor_matrix, memory_matrix, instruction_matrix, input_matrix, output_matrix, proof_stream=None):
        # The trace matrices keep ~10^5 element objects alive; every full garbage collection during (or right after) the proof
        # would walk all of them (measured: 20 ms pauses on a 17 ms proof).  gc.freeze() parks everything that exists now in a
        # permanent generation for the duration of the call; objects made by the proof itself are collected as usual.
        import gc
        from . import debug_checks
        # DEBUG (the reference's switch, brainfuck_stark.py:251-290, table.py:170-176 / 219-234 / 264-284) or BFS_DEBUG=1: every
        # quotient and every term of the combination is interpolated and its degree asserted (debug_checks.py).  The checks need the
        # quotient codewords in HBM, i.e. the path that keeps intermediates; the proof bytes are the same.
        debug = debug_checks.enabled() and not self.keep_intermediates
        if debug:
            self.keep_intermediates = True
        gc.freeze()
        try:
            return self._prove(program, processor_matrix, memory_matrix, instruction_matrix, input_matrix, output_matrix, proof_stream)
        finally:
            gc.unfreeze()
            if debug:
                self.keep_intermediates = False
                self._last = {k: v for k, v in getattr(self, "_last", {}).items()
                              if k in ("challenges", "terminals", "indices", "weights_seed", "quotient_degree_bounds")}

    # ---- the production path: the stages between the Fiat-Shamir points run natively (csrc/prover.cpp), two calls per proof
    native_stages = True            # False: every stage is driven from Python (the path below; what the tests compare the native one with)

    # one native session per proving THREAD (made on a thread's first proof, freed with the thread).  The threading.local itself is made
    # HERE, once, when the class is defined: made lazily, two threads entering their first prove() together could each create one, and the
    # loser's holder -- referenced from nothing but its own frame -- was finalised (bfs_stark_session_free) while the thread still proved
    # with the freed session (round-5 advice).
    _thread_sessions = __import__("threading").local()

    @staticmethod
    def _native_session():
        """the calling thread's bfs_stark session.  A session owns side streams, events and scratch buffers (csrc/prover.cpp), which
        cost far more to make than a small proof takes, so it belongs to the thread, not to the BrainfuckStark object: provers are
        made per claim (running time, program, symbols) and thrown away, threads stay."""
        import weakref
        local = BrainfuckStark._thread_sessions
        holder = getattr(local, "holder", None)
        if holder is None:
            lib = _lib.load()

            class _Holder:
                pass
            holder = local.holder = _Holder()
            holder.session = lib.bfs_stark_session_new()
            weakref.finalize(holder, lib.bfs_stark_session_free, holder.session)
        return holder.session

    @staticmethod
    def _matrix_values(matrix, width):
        """the uint64 array behind a trace matrix of this package's VM (rows x >= width, C order), or None for plain lists of rows"""
        values = getattr(matrix, "values", None)
        if values is None or not isinstance(values, np.ndarray) or values.dtype != np.uint64 or values.ndim != 2:
            return None
        if values.shape[0] != len(matrix) or (values.shape[0] and (values.shape[1] < width or not values.flags.c_contiguous)):
            return None
        return values

    def _prove_native(self, program, matrices, proof_stream):
        """prove() through bfs_stark_commit / bfs_stark_finish.  Returns the proof bytes, or None when this route does not apply (the
        caller then takes the Python path): plain-list matrices, a foreign proof stream, a cooperative proof, test hooks."""
        import os
        from . import salted_merkle as salted_mod, table as table_mod
        from .table import sample_base
        if os.environ.get("BFS_NATIVE_PROVE", "1") == "0" or not self.native_stages:
            return None
        if (self._cooperation is not None or self.keep_intermediates or self.stage_timing or self._row_windows is not None
                or self._shift_tweak is not None):
            return None
        # tables in the order of self.tables: processor, instruction, memory, input, output
        pm, mm, im, inm, om = matrices
        ordered = (pm, im, mm, inm, om)
        values = [BrainfuckStark._matrix_values(m, t.base_width) for m, t in zip(ordered, self.tables)]
        if any(v is None for v in values):
            return None
        if proof_stream is None:
            proof_stream = ProofStream()
        if not hasattr(proof_stream, "_adopt_lazy"):
            return None
        lib, stream = _lib.load(), current_stream()
        xf, n = self.xfield, self.fri.domain.length
        transcript = proof_stream._native()
        if getattr(proof_stream, "_cached", None) is not transcript or (transcript.xfield is not None and transcript.xfield is not xf) or transcript.loaded:
            return None
        if transcript.xfield is None:
            transcript.xfield = xf
        import time
        t_begin = time.perf_counter()
        for table, matrix in zip(self.tables, ordered):
            table.matrix = matrix
        for table, v in zip(self.tables[3:], values[3:]):                       # io_table.py:17-21: length and height follow the symbols
            table.length = v.shape[0]
            table.height = v.shape[0] + table._padding_length(v.shape[0])
        for table, v in zip(self.tables[:3], values[:3]):
            if v.shape[0] + table._padding_length(v.shape[0]) != table.height:
                return None                                                     # (the Python path raises where the reference would)
        # ---- every random draw of prove(), in its order, from the sources the Python path reads (tests replace them module by module)
        rnd = _lib.StarkRandomness()
        keep = []                                                               # buffers the structure points at (ADDRESSES go into it:
        # ctypes.cast(buffer, c_void_p) puts the buffer into its own _objects dictionary -- a reference cycle, and a 24 n-byte salt
        # buffer per commitment then lives until the cyclic collector happens to run: a soak grew by 5 MB per proof that way)
        draw = random_source(urandom)
        count = self.max_degree + 1
        if draw is os.urandom or getattr(draw, "expand_on_device", False):
            keep.append(ctypes.create_string_buffer(draw(32), 32))
            rnd.randomizer_seed = ctypes.addressof(keep[-1])
        else:
            keep.append(np.ascontiguousarray(sample_ext_many(draw(3 * 9 * count), count, 9), dtype=np.uint64))
            rnd.randomizer_limbs = keep[-1].ctypes.data
        tdraw = random_source(table_mod.urandom)

        def draws(source, count):
            """`count` draws of 24 bytes (table.py:125-127), as integers; the operating system's generator is asked once for all of them"""
            if source is os.urandom:
                blob = source(24 * count)
                return [int.from_bytes(blob[24 * i:24 * i + 24], "big") for i in range(count)]
            return [int.from_bytes(source(24), "big") for _ in range(count)]
        base_rand = [v % P_GOLDILOCKS for v in draws(tdraw, sum(t.base_width for t in self.tables[:3] if t.height))]
        keep.append((_u64 * max(len(base_rand), 1))(*base_rand))
        rnd.base_randomizers = ctypes.addressof(keep[-1])

        def salts(field_seed, field_data):
            sdraw = random_source(salted_mod.urandom)
            if sdraw is os.urandom or getattr(sdraw, "expand_on_device", False):
                keep.append(ctypes.create_string_buffer(sdraw(32), 32))
                setattr(rnd, field_seed, ctypes.addressof(keep[-1]))
            else:
                data = sdraw(24 * n)
                keep.append(ctypes.create_string_buffer(data, len(data)))
                setattr(rnd, field_data, ctypes.addressof(keep[-1]))
        salts("base_salt_seed", "base_salts")
        initials = [sample_ext(draw(3 * 8)) for _ in self.permutation_arguments]
        rnd.initials = (_u64 * 6)(*[v for i in initials for v in i])
        mask64 = (1 << 64) - 1              # ExtensionField.sample of 24 bytes: three big-endian 8-byte chunks mod p (extension_field.py:100-111)
        ext_rand = [c % P_GOLDILOCKS for v in draws(tdraw, sum(t.full_width - t.base_width for t in self.tables[:3] if t.height))
                    for c in (v >> 128, (v >> 64) & mask64, v & mask64)]
        keep.append((_u64 * max(len(ext_rand), 1))(*ext_rand))
        rnd.ext_randomizers = ctypes.addressof(keep[-1])
        salts("ext_salt_seed", "ext_salts")

        params = _lib.StarkParams(n.bit_length() - 1, self.expansion_factor, self.num_colinearity_checks, self.security_level,
                                  self.fri.domain.offset.value, self.fri.domain.omega.value, self.max_degree,
                                  (_u64 * 3)(*[t.height for t in self.tables[:3]]))
        tabs = (_lib.StarkTableIn * 5)()
        for slot, v in zip(tabs, values):
            slot.values, slot.rows, slot.row_stride = (v.ctypes.data if v.shape[0] else None), v.shape[0], (v.shape[1] if v.shape[0] else 0)
        session = BrainfuckStark._native_session()
        before = transcript.num_objects()
        out_ch, out_scan, out_io = (_u64 * 33)(), (_u64 * 27)(), (_u64 * 6)()
        ms_a, ms_b = (ctypes.c_double * 5)(), (ctypes.c_double * 5)()
        try:
            _lib.check(lib.bfs_stark_commit(session, transcript.handle, ctypes.byref(params), tabs, ctypes.byref(rnd), out_ch, out_scan, out_io,
                                            ms_a, stream))
            t_commit = time.perf_counter()
            # ---- while the GPU extends the extension columns: terminals, their objects, degree bounds
            challenges = tuple((out_ch[3 * i], out_ch[3 * i + 1], out_ch[3 * i + 2]) for i in range(11))
            scan = [(out_scan[3 * i], out_scan[3 * i + 1], out_scan[3 * i + 2]) for i in range(9)]
            pt, it, mt = self.processor_table, self.instruction_table, self.memory_table
            (pt.instruction_permutation_terminal, pt.memory_permutation_terminal, pt.input_evaluation_terminal,
             pt.output_evaluation_terminal) = scan[0:4]
            it.permutation_terminal, it.evaluation_terminal = scan[4], scan[5]
            mt.permutation_terminal = scan[6]
            self.input_table.evaluation_terminal = (out_io[0], out_io[1], out_io[2])
            self.output_table.evaluation_terminal = (out_io[3], out_io[4], out_io[5])
            ci = values[0][:, 2]
            pt.evaluation_terminal_identities = (pt._identity(None, scan[2], challenges[8], np.nonzero(ci == ord(","))[0] + 1),
                                                 pt._identity(None, scan[3], challenges[9], np.nonzero(ci == ord("."))[0]))
            terminals = self.get_terminals()
            terminal_objects = self._terminal_objects(terminals)
            transcript.scan(terminal_objects)
            handles = (_u64 * 5)(*[transcript.to_native(t) for t in terminal_objects])
            bounds = [t.interpolant_degree() for t in self.tables for _ in range(t.base_width)]
            bounds += [t.interpolant_degree() for t in self.tables for _ in range(t.full_width - t.base_width)]
            quotient_degree_bounds = self._quotient_degree_bounds_cached(challenges, terminals)
            bounds += quotient_degree_bounds
            unit_distances = list(set(table.unit_distance(n) for table in self.tables))
            dist = (_u64 * (1 + len(unit_distances)))(0, *unit_distances)
            out_idx, out_top = (_u64 * max(self.security_level, 1))(), (_u64 * max(self.num_colinearity_checks, 1))()
            wseed = ctypes.create_string_buffer(32)
            t_host = time.perf_counter()
            _lib.check(lib.bfs_stark_finish(session, transcript.handle, handles, (_u64 * 15)(*[v for t in terminals for v in t]),
                                            (_u64 * len(bounds))(*[self.max_degree - b for b in bounds]), len(bounds), transcript._field_id(BrainfuckStark.field), dist, len(dist),
                                            out_idx, wseed, out_top, ms_b, stream))
        except Exception:
            proof_stream._cached = None          # native code may have appended objects the Python list does not have
            raise
        proof_stream._adopt_lazy(transcript, before, transcript.num_objects(), xf)
        t_finish = time.perf_counter()
        proof = proof_stream.serialize()
        self._last = {"challenges": challenges, "terminals": terminals, "indices": [int(v) for v in out_idx[:self.security_level]],
                      "weights_seed": wseed.raw, "quotient_degree_bounds": quotient_degree_bounds}
        self.timing = {"host_prepare": t_commit - t_begin - sum(ms_a) * 1e-3, "pad": ms_a[0] * 1e-3, "randomizer": 0.0, "base_lde": ms_a[1] * 1e-3,
                       "base_tree": ms_a[2] * 1e-3, "extend": ms_a[3] * 1e-3, "ext_lde": ms_a[4] * 1e-3, "host_between_calls": t_host - t_commit,
                       "ext_tree": ms_b[0] * 1e-3, "quotients": 0.0, "combination": ms_b[1] * 1e-3, "combination_tree": ms_b[2] * 1e-3,
                       "openings": ms_b[3] * 1e-3, "fri": ms_b[4] * 1e-3 + (t_finish - t_host - sum(ms_b) * 1e-3),
                       "serialize": time.perf_counter() - t_finish}
        return proof

    _bounds_cache = {}

    def _quotient_degree_bounds_verifier(self, challenges, terminals):
        """the same for verify(): the challenges are Fiat-Shamir outputs, the terminals are the PROVER's, chosen after it has seen the
        challenges.  A terminal made from them (the product of two challenges, say) can cancel a monomial of a terminal constraint while
        looking as sampled as any other value; the reference expands symbolically every time and would then shift that quotient by a
        different amount than the generic bounds say (round-5 advice).  So the terminal constraints -- the only ones the terminals enter
        -- take the exact expansion here (nine small constraints, ~50 us); boundary and transition bounds depend on the challenges
        alone and keep their per-shape memory."""
        out = [b for table in self.tables for b in table.all_quotient_degree_bounds(challenges, terminals, exact_terminals=True)]
        out += [pa.quotient_degree_bound() for pa in self.permutation_arguments]
        return out

    def _quotient_degree_bounds_cached(self, challenges, terminals):
        """all quotient degree bounds of a proof (:203-221) -- Table.all_quotient_degree_bounds of every table, then the permutation
        arguments -- remembered per SHAPE of the inputs.  Which monomials of the composed constraints survive depends on the numeric
        challenges, terminals and parameters only through cancellations (stark_brainfuck_amd/table.py: _degree_bounds); values that
        look sampled (more than 32 significant bits) behave generically except with negligible probability, small ones (zero, the
        `iota^0 = 1` of an IO table without padding, a crafted test value) are part of the key as they are.  The per-table code draws the
        same line but falls back to the exact symbolic expansion whenever ANY value is small -- every proof of a program without input
        does, 160 us of host time between the two native calls."""
        values = list(challenges) + list(terminals) + [p for t in self.tables for p in t.air_params(challenges)]
        sampled = [v for v in values if v[0] >> 32 or v[1] or v[2]]
        if len(set(sampled)) == len(sampled):
            key = (tuple(t.height for t in self.tables), tuple(t.length for t in self.tables[3:]),
                   tuple("s" if (v[0] >> 32 or v[1] or v[2]) else tuple(v) for v in values))
            hit = BrainfuckStark._bounds_cache.get(key)
            if hit is not None:
                return list(hit)
        else:
            key = None
        out = [b for table in self.tables for b in table.all_quotient_degree_bounds(challenges, terminals)]
        out += [pa.quotient_degree_bound() for pa in self.permutation_arguments]
        if key is not None:
            if len(BrainfuckStark._bounds_cache) > 1024:
                BrainfuckStark._bounds_cache.clear()
            BrainfuckStark._bounds_cache[key] = tuple(out)
        return out

    def _terminal_objects(self, terminals):
        """the five terminals as the OBJECTS the reference pushes (:223-224).  The input and output evaluations both start from ONE zero
        object (processor_table.py:340-347) and stay that object when the program never reads / writes; pickle then writes the second
        one as a back-reference.  The running evaluations are sums `evaluation * challenge + lift(symbol)`; the first one is `zero +
        lift(symbol)`, which returns the lifted symbol's polynomial, and from then on the left operand's coefficients --
        BaseFieldElements of the VM's BaseField instance (vm.py:70), not of the extension field's own -- decide the field of every
        result (processor_table.py:390-404, univariate.py:23-35): pickle writes that third BaseField instance out."""
        xf = self.xfield
        terminal_objects = [xf.from_limbs(t) for t in terminals]
        for k, identity in zip((2, 3), self.processor_table.evaluation_terminal_identities):
            limbs = list(terminals[k])
            while limbs and limbs[-1] == 0:
                limbs.pop()
            if limbs and identity is not None and identity[0] == "object":
                terminal_objects[k] = ExtensionFieldElement(Polynomial([identity[1]]), xf)
            elif limbs:
                base = identity[1] if identity is not None else VirtualMachine.field
                terminal_objects[k] = ExtensionFieldElement(Polynomial([BaseFieldElement(v, base) for v in limbs]), xf)
        if not any(terminals[2]) and not any(terminals[3]):
            terminal_objects[3] = terminal_objects[2]
        return terminal_objects

    def _prove(self, program, processor_matrix, memory_matrix, instruction_matrix, input_matrix, output_matrix, proof_stream=None):
        assert len(processor_matrix) + len(program) == len(instruction_matrix)
        proof = self._prove_native(program, (processor_matrix, memory_matrix, instruction_matrix, input_matrix, output_matrix), proof_stream)
        if proof is not None:
            return proof
        lib, stream = _lib.load(), current_stream()
        xf, n = self.xfield, self.fri.domain.length
        log_n = n.bit_length() - 1
        domain = self.fri.domain
        import time
        self.timing = {}
        mark = [time.perf_counter()]
        sync_stages = self.stage_timing
        def lap(name):
            if sync_stages:
                synchronize(stream)
            now = time.perf_counter()
            self.timing[name] = self.timing.get(name, 0.0) + now - mark[0]
            mark[0] = now

        # randomizer polynomial and codeword (:162-167) -- queued FIRST: padding (:143-148) is host work that draws no randomness, so the
        # GPU expands and transforms the randomizer while the host pads (the order of the random draws is the reference's either way)
        count = self.max_degree + 1
        import os
        draw = random_source(urandom)    # the module's urandom, or this context's shared stream (randomness.override)
        if draw is os.urandom or getattr(draw, "expand_on_device", False):
            # production: coefficients expanded on the GPU from 32 bytes of the system's (or the ranks' shared) randomness
            randomizer_polynomial = XArray.empty(count, xf)
            _lib.check(lib.bfs_xfe_sample_fill(draw(32), randomizer_polynomial.ptr, count, count, stream))
        else:                            # a test replaced urandom: the reference's byte stream, `count` draws of 27 bytes
            randomizer_polynomial = XArray.from_numpy(sample_ext_many(draw(3 * 9 * count), count, 9), xf)
        randomizer_codeword = domain.xevaluate(randomizer_polynomial, xf, as_array=True)

        for table, matrix in zip(self.tables, (processor_matrix, instruction_matrix, memory_matrix, input_matrix, output_matrix)):
            table.matrix = matrix
        for table in (self.processor_table, self.memory_table, self.instruction_table, self.input_table, self.output_table):
            table.pad()                                                                      # :143-148
        if proof_stream is None:
            proof_stream = ProofStream()
        lap("pad")           # (includes the randomizer's GPU time where it outlasts the padding)
        lap("randomizer")
        # base codewords of all tables, one commitment to the zipped rows (:169-179)
        lde_tables(self.tables, domain)
        base_degree_bounds = [t.interpolant_degree() for t in self.tables for _ in range(t.base_width)]
        prepared_extension = prepare_extension(self.tables)      # host work (row masks) behind the transform that has just been queued
        lap("base_lde")
        f2 = BrainfuckStark.field

        fetched_base, fetched_ext = {}, {}      # row -> words, filled by the batched gather of the openings

        # the closures below outlive this call inside the trees' lazy leaf lists: they must not hold `self`, or a prover object
        # kept with its trees (keep_intermediates) becomes a reference cycle and its HBM waits for a full garbage collection
        tables = self.tables

        def base_requests(i):
            return [(randomizer_codeword.ptr + 8 * i, 3, randomizer_codeword.stride)] + [(t.base_codewords.ptr + 8 * i, t.base_width, n) for t in tables]

        from .arrays import _fastlist
        internal_field = xf.modulus.coefficients[0].field

        def base_row(i):         # only opened rows are ever read back
            words = fetched_base[i] if i in fetched_base else gather(base_requests(i))
            if _fastlist is not None:     # the same objects, made in C (cpyext/fastlist.c): 16 elements in 3 us instead of 16
                tail = _fastlist.unpack_base(np.ascontiguousarray(words[3:], dtype=np.uint64), BaseFieldElement, f2)
                return tuple([xf.from_limbs([int(v) for v in words[:3]])] + tail)
            return tuple([xf.from_limbs([int(v) for v in words[:3]])] + [BaseFieldElement(int(v), f2) for v in words[3:]])
        base_columns = [(randomizer_codeword.ptr, True, 0)]
        for t in self.tables:
            base_columns += [(t.base_codewords.ptr + 8 * c * n, False, 0) for c in range(t.base_width)]
        base_tree = self._zipped_tree(base_columns, n, base_row)
        proof_stream.push(base_tree.root())
        lap("base_tree")

        # challenges, initials, table extension, terminals (:181-192)
        challenges = tuple(BrainfuckStark._sample_weights(11, proof_stream.prover_fiat_shamir()))
        initials = [sample_ext(draw(3 * 8)) for _ in self.permutation_arguments]
        extend_tables_device(self.tables, challenges, initials, prepared=prepared_extension)   # prefix scans on the trace columns lde() left in HBM
        terminals = self.get_terminals()
        lap("extend")

        # extension codewords and their commitment (:194-201)
        lde_tables(self.tables, domain, extension=True)
        extension_degree_bounds = [t.interpolant_degree() for t in self.tables for _ in range(t.full_width - t.base_width)]
        early_quotient_bounds = None
        if not self.keep_intermediates:
            # the quotient degree bounds (:203-221) are host work on challenges and terminals: done here, while the GPU runs the coset
            # transform of the extension columns that lde_tables has just queued
            early_quotient_bounds = [b for table in self.tables for b in table.all_quotient_degree_bounds(challenges, terminals)]
            early_quotient_bounds += [pa.quotient_degree_bound() for pa in self.permutation_arguments]
        lap("ext_lde")
        num_ext_columns = sum(t.full_width - t.base_width for t in self.tables)
        moduli = [m for t in self.tables for m in t.ext_sharing_moduli(n)]
        internal = xf.modulus.coefficients[0].field
        shared = [dict() for _ in moduli]          # per column: i mod modulus -> the coefficient objects of that class

        def ext_requests(i):
            return [(t.ext_codewords.ptr + 8 * i, 3 * (t.full_width - t.base_width), n) for t in tables]

        plain_columns = [c for c in range(num_ext_columns) if moduli[c] is None]

        def ext_row(i):
            words = fetched_ext[i] if i in fetched_ext else gather(ext_requests(i))
            row = []
            made = None
            if _fastlist is not None and plain_columns:
                # the elements without shared coefficient objects, all at once: limb planes (3, k) -> k ExtensionFieldElements
                soa = np.ascontiguousarray(np.asarray(words, dtype=np.uint64).reshape(num_ext_columns, 3)[plain_columns].T)
                made = dict(zip(plain_columns, _fastlist.unpack_ext(soa, ExtensionFieldElement, Polynomial, BaseFieldElement, xf, internal_field)))
            for c in range(num_ext_columns):
                if made is not None and c in made:
                    row.append(made[c])
                    continue
                limbs = [int(v) for v in words[3 * c:3 * c + 3]]
                if moduli[c] is None:
                    row.append(xf.from_limbs(limbs))
                    continue
                while limbs and limbs[-1] == 0:
                    limbs.pop()
                objs = shared[c].setdefault(i % moduli[c], [BaseFieldElement(v, internal) for v in limbs])
                e = ExtensionFieldElement(Polynomial(objs), xf)
                e.shares_coefficients = True
                row.append(e)
            return tuple(row)
        ext_columns = []
        for t in self.tables:
            ext_columns += [(t.ext_codewords.ptr + 8 * 3 * c * n, True, 0) for c in range(t.full_width - t.base_width)]
        extension_tree = self._zipped_tree(ext_columns, n, ext_row)
        proof_stream.push(extension_tree.root())
        lap("ext_tree")

        # quotients (:203-221)
        # keep_intermediates (tests): the quotient codewords are written out and summed by bfs_combination, as the reference
        # does; otherwise they only ever exist in registers (bfs_air_combine below).  Same field elements either way.
        quotient_buffers, quotient_degree_bounds = [], []
        if early_quotient_bounds is not None:
            quotient_degree_bounds = early_quotient_bounds
        else:
            for table in self.tables:
                quotient_buffers.append((table.all_quotients(domain, None, challenges, terminals), table.num_quotients()))
                quotient_degree_bounds += table.all_quotient_degree_bounds(challenges, terminals)
            for pa in self.permutation_arguments:
                quotient_buffers.append((pa.quotient(domain), 1))
                quotient_degree_bounds.append(pa.quotient_degree_bound())
            from . import debug_checks
            if debug_checks.enabled():
                # the reference checks each table's quotients inside all_quotients (before the terminals are pushed) ...
                supports = [debug_checks.check_table_quotients(table, buf, n, domain.omega.value)
                            for table, (buf, _) in zip(self.tables, quotient_buffers)]
                supports += [debug_checks.support(buf.ptr, 3, count, n, domain.omega.value) for buf, count in quotient_buffers[len(self.tables):]]
                # ... and the terms of the combination while it assembles them (after the weights are sampled; nothing in between
                # depends on the outcome, so both sets run here)
                debug_checks.check_terms(self, n, domain.omega.value, base_degree_bounds, extension_degree_bounds, supports, quotient_degree_bounds)

        lap("quotients")
        terminal_objects = self._terminal_objects(terminals)          # :223-224
        for t in terminal_objects:
            proof_stream.push(t)

        # weights of the non-linear combination (:226-243)
        num_base = sum(t.base_width for t in self.tables)
        num_ext = sum(t.full_width - t.base_width for t in self.tables)
        num_quot = len(quotient_degree_bounds)
        weights_seed = proof_stream.prover_fiat_shamir()
        weight_array = BrainfuckStark._sample_weights(1 + 2 * (num_base + num_ext + num_quot), weights_seed, as_array=True)
        weight0 = tuple(int(v) for v in weight_array[0])

        # terms in the order of the reference's `terms` list (:245-293): base, extension, quotient codewords; term s has the
        # weights 1 + 2s, 2 + 2s and is shifted to the common degree bound.  One row of seven words per term (bfs_comb_weight).
        bounds = base_degree_bounds + extension_degree_bounds + quotient_degree_bounds
        assert 1 + 2 * len(bounds) == len(weight_array)
        terms = np.empty((len(bounds), 7), dtype=np.uint64)
        terms[:, 0:3] = weight_array[1::2]
        terms[:, 3:6] = weight_array[2::2]
        terms[:, 6] = [self.max_degree - bound for bound in bounds]
        if self._shift_tweak is not None:
            terms[:, 6] = self._shift_tweak(terms[:, 6])

        def term_of(s):
            return tuple(int(v) for v in terms[s, 0:3]), tuple(int(v) for v in terms[s, 3:6]), int(terms[s, 6])
        combination = XArray.empty(n, xf)
        if self.keep_intermediates:
            sources = []
            for t in self.tables:
                for c in range(t.base_width):
                    sources.append((t.base_codewords.ptr + 8 * c * n, 0))
            for t in self.tables:
                for c in range(t.full_width - t.base_width):
                    sources.append((t.ext_codewords.ptr + 8 * 3 * c * n, 1))
            for buf, count in quotient_buffers:
                for q in range(count):
                    sources.append((buf.ptr + 8 * 3 * q * n, 1))
            assert len(sources) == len(bounds)
            srcs = (_lib.CombSource * len(sources))()
            for s, (ptr, is_ext) in enumerate(sources):
                wa, wb, shift = term_of(s)
                srcs[s].ptr, srcs[s].is_ext, srcs[s].shift = ptr, is_ext, shift
                srcs[s].wa = (_u64 * 3)(*wa)
                srcs[s].wb = (_u64 * 3)(*wb)
            _lib.check(lib.bfs_combination(srcs, len(sources), randomizer_codeword.ptr, (_u64 * 3)(*weight0), combination.ptr,
                                           log_n, domain.offset.value, domain.omega.value, stream))
        else:
            # a cooperative proof (cooperate()): the stage is pointwise, every rank holds all codewords (a row's neighbour at
            # unit_distance comes from the rank's own copy), so each rank does its own rows and the ranks all-gather the combination.
            # _row_windows (tests): the same row-window entry points on one GPU, the domain cut into arbitrary pieces.
            rows = None
            if self._cooperation is not None:
                from .shard import row_range
                rows = row_range(n, self._cooperation[0], self._cooperation[1])
            windows = [rows]
            if rows is None and self._row_windows is not None:
                windows = list(self._row_windows)
                ends = [first + count for first, count in windows]
                assert [first for first, _ in windows] == [0] + ends[:-1] and ends[-1] == n, "the windows must tile the domain in order"
            for window in windows:
                inverse_buffer, inverses = zerofier_inverses(self.tables, domain, rows=window)      # all zerofier denominators, one inversion per point
                base_at = ext_at = 0
                quot_at = num_base + num_ext
                for k, t in enumerate(self.tables):
                    bw, xw, nq = t.base_width, t.full_width - t.base_width, t.num_quotients()
                    mine = np.concatenate([terms[base_at:base_at + bw], terms[num_base + ext_at:num_base + ext_at + xw], terms[quot_at:quot_at + nq]])
                    t.combine_into(domain, challenges, terminals, mine, combination,
                                   randomizer=randomizer_codeword if k == 0 else None, randomizer_weight=weight0, inverses=inverses[t], rows=window)
                    base_at, ext_at, quot_at = base_at + bw, ext_at + xw, quot_at + nq
                for pa in self.permutation_arguments:
                    pa.combine_into(domain, term_of(quot_at), combination, inv_x_minus_1=inverses[self.tables[0]][0], rows=window)
                    quot_at += 1
                assert quot_at == len(terms)
                inverse_buffer.free()
            if rows is not None:
                from .shard import all_gather_rows
                world_size, rank, group, device = self._cooperation
                all_gather_rows(combination.ptr, n, 3, combination.stride, world_size, rank, group=group, device=device, stream=stream)

        if not self.keep_intermediates:
            BrainfuckStark._release(randomizer_polynomial, *[buf for buf, _ in quotient_buffers])
            quotient_buffers = []
        lap("combination")
        # commitment to the combination codeword, openings (:300-333)
        combination_tree = Merkle(combination)
        proof_stream.push(combination_tree.root())
        lap("combination_tree")          # (GPU work: 2^22 extension leaves are 1.1 ms; the round-3 breakdown counted it under "openings")
        indices = BrainfuckStark.sample_indices(self.security_level, proof_stream.prover_fiat_shamir(), n)
        unit_distances = list(set(table.unit_distance(n) for table in self.tables))
        known = self._openings_native(proof_stream, base_tree, extension_tree, combination, combination_tree, base_requests(0),
                                      ext_requests(0), moduli, indices, unit_distances, n, f2, xf, lib, stream)
        if known is None:
            known = self._openings_python(proof_stream, base_tree, extension_tree, combination, combination_tree, base_requests,
                                          ext_requests, fetched_base, fetched_ext, indices, unit_distances, n, xf)

        if not self.keep_intermediates:
            BrainfuckStark._release(base_tree, extension_tree, randomizer_codeword, *self.tables)
            base_tree = extension_tree = None
        lap("openings")
        # low-degree test of the combination codeword (:335-336)
        self.fri.prove(combination, proof_stream, known_leafs=known, round0_tree=combination_tree)   # round 0 commits to this very tree
        self._last = {"challenges": challenges, "terminals": terminals, "indices": indices, "weights_seed": weights_seed,
                      "quotient_degree_bounds": quotient_degree_bounds}
        if self.keep_intermediates:
            self._last.update({"base_tree": base_tree, "extension_tree": extension_tree, "combination_tree": combination_tree,
                               "quotient_buffers": quotient_buffers, "combination": combination,
                               "randomizer_codeword": randomizer_codeword})
        else:
            BrainfuckStark._release(combination, combination_tree)
        lap("fri")
        proof = proof_stream.serialize()
        lap("serialize")
        return proof

    # ------------------------------------------------------------------------------------------------------------
    def verify(self, proof, proof_stream=None):
        """brainfuck_stark.py:343-579 -- host only, like the reference's verifier: Merkle paths of the opened rows, the
        non-linear combination recomputed from the opened rows (constraints evaluated through air.evaluate), FRI, and the
        terminals against the public input, output and program."""
        from .air import X0, xadd, xmul, xscale
        P = air.P
        if proof_stream is None and self.native_stages:
            verdict = self._verify_native(proof)
            if verdict is not None:
                return verdict
        if proof_stream is None:
            proof_stream = ProofStream()
        proof_stream = proof_stream.deserialize(proof)
        if hasattr(proof_stream, "pickle_of"):
            # leaf preimages (pickle.dumps of an opened row / element) come from the native copy of the stream while this call runs
            from .merkle import leaf_pickle_source
            token = leaf_pickle_source.set(proof_stream.pickle_of)
            try:
                return self._verify_checked(proof_stream)
            finally:
                leaf_pickle_source.reset(token)
        return self._verify_checked(proof_stream)

    def _verify_native(self, proof):
        """verify() on the native object graph of the proof (csrc/verifier.cpp: bfs_stark_verify_begin / _finish): the same checks in the same
        order as _verify_stream / Fri.verify below, without a Python object per pulled item.  Returns True / False, raises the reference's
        AssertionError -- or returns None when this route does not apply (the bytes are not something the native reader takes, or the stream
        holds an object the native checks do not model): the Python verifier below then decides, as the reference would."""
        import os
        if os.environ.get("BFS_NATIVE_VERIFY", "1") == "0":
            return None
        from .ip import NativeTranscript
        try:
            data = bytes(proof)
        except TypeError:
            return None
        t = NativeTranscript.from_bytes(data)
        if t is None:
            return None
        lib = _lib.load()
        n = self.fri.domain.length
        unit_distances = list(set(table.unit_distance(n) for table in self.tables))
        if len(unit_distances) > 8:
            return None
        words = (_u64 * max(len(self.program), 1))(*[w.value if hasattr(w, "value") else int(w) for w in self.program])
        ins = (_u64 * max(len(self.input_symbols), 1))(*[ord(c) for c in self.input_symbols])
        outs = (_u64 * max(len(self.output_symbols), 1))(*[ord(c) for c in self.output_symbols])
        params = _lib.StarkVerifyParams()
        params.log_n, params.expansion_factor = n.bit_length() - 1, self.expansion_factor
        params.num_colinearity_checks, params.security_level = self.num_colinearity_checks, self.security_level
        params.offset, params.omega = self.fri.domain.offset.value, self.fri.domain.omega.value
        params.heights = (_u64 * 5)(*[t_.height for t_ in self.tables])
        params.lengths = (_u64 * 5)(*[t_.length for t_ in self.tables])
        params.omicrons = (_u64 * 5)(*[t_.omicron.value for t_ in self.tables])
        params.num_distances = len(unit_distances)
        params.distances = (_u64 * 8)(*(unit_distances + [0] * (8 - len(unit_distances))))
        params.program, params.program_len = ctypes.addressof(words), len(self.program)
        params.input, params.n_input = ctypes.addressof(ins), len(self.input_symbols)
        params.output, params.n_output = ctypes.addressof(outs), len(self.output_symbols)
        out_ch, out_tm, verdict = (_u64 * 33)(), (_u64 * 15)(), ctypes.c_int(3)
        _lib.check(lib.bfs_stark_verify_begin(t.handle, ctypes.byref(params), out_ch, out_tm, ctypes.byref(verdict)))
        if verdict.value == 2:
            raise AssertionError(lib.bfs_last_error().decode("utf-8", "replace"))
        if verdict.value != 1:
            return None if verdict.value == 3 else False
        challenges = tuple((out_ch[3 * i], out_ch[3 * i + 1], out_ch[3 * i + 2]) for i in range(11))
        terminals = [(out_tm[3 * i], out_tm[3 * i + 1], out_tm[3 * i + 2]) for i in range(5)]
        bounds = [t_.interpolant_degree() for t_ in self.tables for _ in range(t_.base_width)]
        bounds += [t_.interpolant_degree() for t_ in self.tables for _ in range(t_.full_width - t_.base_width)]
        bounds += self._quotient_degree_bounds_verifier(challenges, terminals)
        shifts = (_u64 * len(bounds))(*[self.max_degree - b for b in bounds])
        _lib.check(lib.bfs_stark_verify_finish(t.handle, ctypes.byref(params), shifts, len(bounds), ctypes.byref(verdict)))
        if verdict.value == 2:
            raise AssertionError(lib.bfs_last_error().decode("utf-8", "replace"))
        if verdict.value == 3:
            return None
        return verdict.value == 1

    def _verify_checked(self, proof_stream):
        try:
            return self._verify_stream(proof_stream)
        except _MalformedProof:
            return False

    def _verify_stream(self, proof_stream):
        from .air import X0, xadd, xmul, xscale
        P = air.P
        n = self.fri.domain.length
        offset, omega = self.fri.domain.offset.value, self.fri.domain.omega.value

        def limbs(e):
            """value of an element read from the proof, as canonical residues: the prover chooses the representation (any Python int
            unpickles), the reference reduces in every operation (algebra.py:89-99) and so sees v mod p -- and so must every
            consumer here: bfs_air_evaluate's host arithmetic assumes canonical operands and ctypes truncates above 2^64
            (round-4 advice).  An extension element with more than three coefficients is not an element: the proof is refused."""
            if hasattr(e, "limbs"):
                c = e.limbs()
                if len(c) != 3:
                    raise _MalformedProof("extension element with %d coefficients" % len(c))
                return (c[0] % P, c[1] % P, c[2] % P)
            return (e.value % P, 0, 0)

        base_root = proof_stream.pull()
        challenges = tuple(BrainfuckStark._sample_weights(11, proof_stream.verifier_fiat_shamir()))
        extension_root = proof_stream.pull()
        terminal_objects = [proof_stream.pull() for _ in range(5)]
        terminals = [limbs(t) for t in terminal_objects]
        # ... and as STORED, for the three evaluation arguments at the end: the reference compares the pulled object with a computed
        # element through Polynomial.__eq__ / BaseFieldElement.__eq__, i.e. the coefficient values as they were pickled (round-5 advice)
        stored_terminals = [tuple(t.limbs()) if hasattr(t, "limbs") else (t.value, 0, 0) for t in terminal_objects]

        base_degree_bounds = [t.interpolant_degree() for t in self.tables for _ in range(t.base_width)]
        extension_degree_bounds = [t.interpolant_degree() for t in self.tables for _ in range(t.full_width - t.base_width)]
        num_base = sum(t.base_width for t in self.tables)
        num_ext = sum(t.full_width - t.base_width for t in self.tables)
        num_quot = sum(t.num_quotients(challenges, terminals) for t in self.tables)
        weights = BrainfuckStark._sample_weights(1 + 2 * (num_base + num_ext + num_quot + len(self.permutation_arguments)),
                                                 proof_stream.verifier_fiat_shamir(), as_array=True)
        weights = np.ascontiguousarray(weights, dtype=np.uint64)      # (number, 3); stays alive for weights_raw below
        weights_raw = weights.ctypes.data_as(ctypes.POINTER(_u64))
        combination_root = proof_stream.pull()
        indices = BrainfuckStark.sample_indices(self.security_level, proof_stream.verifier_fiat_shamir(), n)
        unit_distances = list(set(table.unit_distance(n) for table in self.tables))

        rows = {}
        for index in indices:
            for distance in [0] + unit_distances:
                idx = (index + distance) % n
                element = proof_stream.pull()
                salt, path = proof_stream.pull()
                assert SaltedMerkle.verify(base_root, idx, salt, path, element), "salted base tree verify must succeed for base codewords"
                row = [limbs(e) for e in element]
                element = proof_stream.pull()
                salt, path = proof_stream.pull()
                assert SaltedMerkle.verify(extension_root, idx, salt, path, element), \
                    "salted base tree verify must succeed for extension codewords"
                rows[idx] = row + [limbs(e) for e in element]

        quotient_bounds = {t: (t.boundary_quotient_degree_bounds(challenges), t.transition_quotient_degree_bounds(challenges),
                               t.terminal_quotient_degree_bounds(challenges, terminals, exact=True)) for t in self.tables}      # (exact: see _quotient_degree_bounds_verifier)
        for index in indices:
            x = offset * pow(omega, index, P) % P

            powers = {}                 # x^shift for the few distinct shifts of a proof (151 terms share ~20 degree bounds)

            def shifted(value, bound):
                f = powers.get(bound)
                if f is None:
                    f = powers[bound] = pow(x, self.max_degree - bound, P)
                return xscale(value, f)
            row = rows[index]
            terms = [row[0]]
            for i in range(num_base):
                terms += [row[1 + i], shifted(row[1 + i], base_degree_bounds[i])]
            ext_offset = 1 + num_base
            for i in range(num_ext):
                terms += [row[ext_offset + i], shifted(row[ext_offset + i], extension_degree_bounds[i])]

            # the rows of every table: base columns, then its extension columns
            points, next_points = [], []
            b, e = 1, ext_offset
            for table in self.tables:
                xw = table.full_width - table.base_width
                nrow = rows[(index + table.unit_distance(n)) % n]
                points.append(row[b:b + table.base_width] + row[e:e + xw])
                next_points.append(nrow[b:b + table.base_width] + nrow[e:e + xw])
                b, e = b + table.base_width, e + xw

            boundary_inverse = pow((x - 1) % P, P - 2, P)
            for table, point, next_point in zip(self.tables, points, next_points):
                bb, tb, zb = quotient_bounds[table]
                omicron_inverse = pow(table.omicron.value, P - 2, P)
                boundary_values, transition_values, terminal_values = table.evaluate_all_constraints(point, next_point, challenges, terminals)
                for value, bound in zip(boundary_values, bb):
                    q = xscale(value, boundary_inverse)
                    terms += [q, shifted(q, bound)]
                if table.height == 0:
                    transition_factor = 0
                else:
                    transition_factor = (x - omicron_inverse) * pow((pow(x, table.height, P) - 1) % P, P - 2, P) % P
                for value, bound in zip(transition_values, tb):
                    q = xscale(value, transition_factor)
                    terms += [q, shifted(q, bound)]
                terminal_inverse = pow((x - omicron_inverse) % P, P - 2, P)
                for value, bound in zip(terminal_values, zb):
                    q = xscale(value, terminal_inverse)
                    terms += [q, shifted(q, bound)]
            for arg in self.permutation_arguments:
                q = xscale(arg.evaluate_difference(points), boundary_inverse)
                terms += [q, shifted(q, arg.quotient_degree_bound())]
            assert len(terms) == len(weights), f"length of terms ({len(terms)}) must be equal to length of weights ({len(weights)})"
            flat = (_u64 * (3 * len(terms)))(*[v for t in terms for v in t])          # bfs_xfe_inner_product: 303 products natively
            got = (_u64 * 3)()
            _lib.check(_lib.load().bfs_xfe_inner_product(weights_raw, flat, len(terms), got))
            inner_product = (got[0], got[1], got[2])

            combination_leaf = proof_stream.pull()
            combination_path = proof_stream.pull()
            if not Merkle.verify(combination_root, index, combination_path, combination_leaf):
                return False
            # brainfuck_stark.py:567 compares the leaf OBJECT with the inner product (coefficient values as stored, algebra.py:36):
            # a leaf whose coefficients are not canonical residues is unequal there, and here
            if not hasattr(combination_leaf, "limbs") or tuple(combination_leaf.limbs()) != inner_product:
                return False

        verdict = self.fri.verify(proof_stream, combination_root)
        for ea in self.evaluation_arguments:
            verdict = verdict and tuple(ea.select_terminal(stored_terminals)) == tuple(ea.compute_terminal(challenges))
        return bool(verdict)
