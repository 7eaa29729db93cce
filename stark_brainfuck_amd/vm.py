"""Brainfuck virtual machine and execution-trace recorder -- host-side mirror of the reference's `vm.py`
(/root/reference/code/vm.py:69-306): `VirtualMachine.compile / run / simulate / execute`, same return values.
Scalar control code (a few thousand rows for "Hello World"): it stays on the host, like the reference's.
Matrices are lists of rows of BaseFieldElement, as in the reference, so they can be handed to either prover.
"""
from .algebra import BaseField, BaseFieldElement

P = (1 << 64) - (1 << 32) + 1


class TraceMatrix(list):
    """a matrix in the reference's format (list of rows of BaseFieldElement) that also carries the same values as a uint64
    array (`values`, rows x columns), so that the tables need not unbox hundreds of thousands of elements again"""
    values = None


class LazyTraceMatrix:
    """A matrix in the reference's format -- a sequence of rows of BaseFieldElement -- over a uint64 array (`values`, rows x
    columns) that makes the element objects of a row only when the row is asked for, and then keeps them (a row read twice is
    the same objects twice).  One column may carry object ids: entries with equal ids ARE the same object, across rows and across
    matrices that share the registry -- the reference's memory cells hold element objects, and the memory-value column, the input
    and the output matrix all point at them (vm.py:266-292); pickle memoises by identity, so this shows in proofs."""

    def __init__(self, values, field, id_column=None, ids=None, registry=None):
        self.values, self._field, self._rows = values, field, {}
        self._id_column, self._ids, self._registry = id_column, ids, registry

    def __len__(self):
        return self.values.shape[0]

    def _row(self, r):
        row = self._rows.get(r)
        if row is None:
            field = self._field
            row = [BaseFieldElement(int(v), field) for v in self.values[r]]
            if self._id_column is not None:
                key = int(self._ids[r])
                row[self._id_column] = self._registry.setdefault(key, row[self._id_column])
            self._rows[r] = row
        return row

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self._row(r) for r in range(*index.indices(len(self)))]
        if index < 0:
            index += len(self)
        if not 0 <= index < len(self):
            raise IndexError("matrix row out of range")
        return self._row(index)

    def __iter__(self):
        return (self._row(r) for r in range(len(self)))

    def __eq__(self, other):
        return len(self) == len(other) and all(a == b for a, b in zip(self, other))


def _matrix(rows, width, field, objects=None):
    import numpy as np
    m = TraceMatrix(objects if objects is not None else [[BaseFieldElement(v, field) for v in r] for r in rows])
    m.values = np.array(rows, dtype=np.uint64).reshape(len(rows), width)
    return m


class VirtualMachine:
    field = BaseField.main()
    # not in the reference: every entry point that runs a program stops with an AssertionError after this many cycles unless
    # the caller asks for another limit (`-[-]` counts down from p - 1 and a trace costs ~130 bytes per cycle natively,
    # several hundred as element objects).  = BFS_VM_DEFAULT_MAX_CYCLES of include/bfstark.h
    DEFAULT_MAX_CYCLES = 1 << 24
    UNLIMITED = float("inf")        # max_cycles = VirtualMachine.UNLIMITED: run on like the reference does (no cap at all)

    @staticmethod
    def _cycle_limit(max_cycles):
        """None and 0 mean DEFAULT_MAX_CYCLES (0 as in bfs_vm_trace_new); UNLIMITED or anything >= 2^64 - 1 means no limit"""
        if max_cycles is None or max_cycles == 0:
            return VirtualMachine.DEFAULT_MAX_CYCLES
        assert max_cycles > 0, "max_cycles must be positive (0 / None: the default, VirtualMachine.UNLIMITED: none)"
        return (1 << 64) - 1 if max_cycles >= (1 << 64) - 1 else int(max_cycles)

    @staticmethod
    def execute(brainfuck_code):
        program = VirtualMachine.compile(brainfuck_code)
        return VirtualMachine.run(program)

    @staticmethod
    def compile(brainfuck_code):
        """vm.py:78-105: one word per symbol; `[` and `]` are followed by the jump target (the index just behind the
        matching bracket's target slot)."""
        field = VirtualMachine.field
        words, stack = [], []
        for symbol in brainfuck_code:
            words.append(ord(symbol))
            if symbol == "[":
                words.append(0)
                stack.append(len(words) - 1)
            elif symbol == "]":
                words.append(stack[-1] + 1)
                words[stack[-1]] = len(words)
                stack.pop()
        return [BaseFieldElement(w, field) for w in words]

    @staticmethod
    def _words(program):
        return [w.value if hasattr(w, "value") else int(w) for w in program]

    @staticmethod
    def run(program, input_data=[], max_cycles=None):
        """vm.py:107-165 -> (running_time, input_data, output_data).  Input symbols that are not supplied cannot be
        read from a terminal here: running out of input is an error.  max_cycles: see DEFAULT_MAX_CYCLES."""
        prog = VirtualMachine._words(program)
        limit = VirtualMachine._cycle_limit(max_cycles)
        ip, mp, memory = 0, 0, {}
        output_data, input_data, input_counter = [], list(input_data), 0
        running_time = 1
        while ip < len(prog):
            w = prog[ip]
            if w == ord("["):
                ip = prog[ip + 1] if memory.get(mp, 0) == 0 else ip + 2
            elif w == ord("]"):
                ip = prog[ip + 1] if memory.get(mp, 0) != 0 else ip + 2
            elif w == ord("<"):
                ip, mp = ip + 1, (mp - 1) % P
            elif w == ord(">"):
                ip, mp = ip + 1, (mp + 1) % P
            elif w == ord("+"):
                ip, memory[mp] = ip + 1, (memory.get(mp, 0) + 1) % P
            elif w == ord("-"):
                ip, memory[mp] = ip + 1, (memory.get(mp, 0) - 1) % P
            elif w == ord("."):
                ip += 1
                output_data += chr(memory[mp] % 256)
            elif w == ord(","):
                ip += 1
                assert input_counter < len(input_data), "program reads more input symbols than were supplied"
                memory[mp] = ord(input_data[input_counter])
                input_counter += 1
            else:
                assert False, f"unrecognized instruction at {ip}: {w}"
            running_time += 1
            assert running_time <= limit, f"program runs for more than {limit} cycles"
        return running_time, input_data, output_data

    @staticmethod
    def simulate(program, input_data=[], max_cycles=None):
        """vm.py:172-306 -> (processor_matrix, memory_matrix, instruction_matrix, input_matrix, output_matrix).
        max_cycles (not in the reference; default DEFAULT_MAX_CYCLES): stop with an AssertionError instead of running on --
        `-[-]` counts down from p - 1 and the trace would grow until the process is killed.
        The machine runs natively (bfs_vm_trace_new, csrc/vm.cpp: 37 000 cycles in milliseconds instead of seconds of element
        construction); the matrices are LazyTraceMatrix objects over integer arrays.  `simulate_objects` is the direct
        restatement that builds every element, kept as the cross-check."""
        import ctypes
        import numpy as np
        from . import _lib
        lib = _lib.load()
        field = VirtualMachine.field
        words = VirtualMachine._words(program)
        prog = (ctypes.c_uint64 * len(words))(*words)
        symbols = [ord(c) if isinstance(c, str) else int(c) for c in input_data]
        inp = (ctypes.c_uint32 * max(len(symbols), 1))(*symbols)
        handle = ctypes.c_void_p()
        limit = VirtualMachine._cycle_limit(max_cycles)
        rc = lib.bfs_vm_trace_new(prog, len(words), inp, len(symbols), limit, ctypes.byref(handle))
        if rc:
            message = lib.bfs_last_error().decode("utf-8", "replace")
            assert False, message          # the reference's asserts: unrecognized instruction / input exhausted
        try:
            def part(which, width):
                n = ctypes.c_size_t()
                _lib.check(lib.bfs_vm_trace_size(handle, which, ctypes.byref(n)))
                out = np.empty(n.value, dtype=np.uint64)
                if n.value:
                    _lib.check(lib.bfs_vm_trace_copy(handle, which, out.ctypes.data))
                return out.reshape(n.value // width, width)
            registry = {}
            processor = LazyTraceMatrix(part(0, 7), field, 5, part(5, 1).reshape(-1), registry)
            memory = LazyTraceMatrix(part(1, 4), field)
            instruction = LazyTraceMatrix(part(2, 3), field)
            inputs = LazyTraceMatrix(part(3, 1), field, 0, part(6, 1).reshape(-1), registry)
            outputs = LazyTraceMatrix(part(4, 1), field, 0, part(7, 1).reshape(-1), registry)
        finally:
            lib.bfs_vm_trace_free(handle)
        return processor, memory, instruction, inputs, outputs

    @staticmethod
    def simulate_objects(program, input_data=[], max_cycles=1 << 20):
        """the same five matrices built the reference's way, every element an object (vm.py:172-306); cycle limit as in
        `simulate`, lower by default because every cycle builds seven element objects"""
        from .memory_table import MemoryTable
        field = VirtualMachine.field
        prog = VirtualMachine._words(program)
        n = len(prog)
        clk = ip = mp = mvi = 0
        ci = prog[0]
        ni = prog[1] if n > 1 else 0
        # Memory cells hold element OBJECTS, and the memory-value register is whatever object sits in the current cell, as
        # in the reference (vm.py:266-292): the object identity of these entries reaches the proof through the first term of
        # the input / output running evaluations (processor_table.py:390-404), and pickle memoises by identity.
        zero = BaseFieldElement(0, field)
        memory, input_counter = {}, 0
        mv = BaseFieldElement(0, field)
        processor, inputs, outputs = [], [], []
        instruction = [[i, prog[i], prog[i + 1]] for i in range(n - 1)] + [[n - 1, prog[-1], 0]]

        processor_values = []

        def row():
            processor_values.append([clk, ip, ci, ni, mp, mv.value, mvi])
            return [BaseFieldElement(clk, field), BaseFieldElement(ip, field), BaseFieldElement(ci, field), BaseFieldElement(ni, field),
                    BaseFieldElement(mp, field), mv, BaseFieldElement(mvi, field)]
        while ip < n:
            processor.append(row())
            instruction.append([ip, ci, ni])
            if ci == ord("["):
                ip = prog[ip + 1] if mv.value == 0 else (ip + 2) % P
            elif ci == ord("]"):
                ip = prog[ip + 1] if mv.value != 0 else (ip + 2) % P
            elif ci == ord("<"):
                ip, mp = ip + 1, (mp - 1) % P
            elif ci == ord(">"):
                ip, mp = ip + 1, (mp + 1) % P
            elif ci == ord("+"):
                ip, memory[mp] = ip + 1, BaseFieldElement((memory.get(mp, zero).value + 1) % P, field)
            elif ci == ord("-"):
                ip, memory[mp] = ip + 1, BaseFieldElement((memory.get(mp, zero).value - 1) % P, field)
            elif ci == ord("."):
                ip += 1
                outputs.append([memory.get(mp, zero)])
            elif ci == ord(","):
                ip += 1
                assert input_counter < len(input_data), "program reads more input symbols than were supplied"
                memory[mp] = BaseFieldElement(ord(input_data[input_counter]), field)
                input_counter += 1
                inputs.append([memory[mp]])
            else:
                assert False, f"unrecognized instruction at {ip}: '{chr(ci)}'"
            clk += 1
            assert clk <= max_cycles, f"program runs for more than {max_cycles} cycles"
            ci = prog[ip] if ip < n else 0
            ni = prog[ip + 1] if ip < n - 1 else 0
            mv = memory.get(mp, zero)
            mvi = pow(mv.value, P - 2, P) if mv.value else 0
        processor.append(row())
        instruction.append([ip, ci, ni])
        instruction.sort(key=lambda r: r[0])          # stable, by address (vm.py:302)
        processor = _matrix(processor_values, 7, field, processor)
        memory = MemoryTable.derive_matrix(processor)
        return (processor, memory, _matrix(instruction, 3, field),
                _matrix([[r[0].value] for r in inputs], 1, field, inputs), _matrix([[r[0].value] for r in outputs], 1, field, outputs))

    @staticmethod
    def num_challenges():
        return 11
