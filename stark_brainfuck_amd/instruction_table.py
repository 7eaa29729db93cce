"""Instruction table -- mirror of the reference's `instruction_table.py` (/root/reference/code/instruction_table.py):
padding (:19-25) and `extend` (:167-231).  Constraints: air.InstructionAir."""
import numpy as np

from . import air
from .air import X0
from .table import Table


class InstructionTable(Table):
    address, current_instruction, next_instruction, permutation, evaluation = range(5)
    air = air.TABLE_AIRS[1]
    table_index = 1

    def __init__(self, field, length, num_randomizers, generator, order):
        super().__init__(field, 3, 5, length, num_randomizers, generator, order)

    def pad(self):
        m = self.base_array()
        pad = np.zeros((3, self._padding_length(m.shape[1])), dtype=np.uint64)
        pad[0] = m[0, -1] if m.shape[1] else 0                         # the last address repeats (instruction_table.py:19-25)
        self._pad_to(pad)

    def extend(self, all_challenges, all_initials):
        """instruction_table.py:167-231"""
        a, b, c, d, e, f, alpha, beta, gamma, delta, eta = all_challenges
        m = self.base_array()
        addr, ci, ni = m[0], m[1], m[2]
        same = np.concatenate([[False], addr[1:] == addr[:-1]]) if len(addr) else np.zeros(0, dtype=bool)
        # the running product absorbs a row when it is not padding and repeats the previous row's address (:197-205);
        # the running evaluation absorbs the first row of every address (:209-214); both are recorded AFTER the row's update
        f_perm = self.scan_async(0, [addr, ci, ni], (ci != 0) & same, [alpha, a, b, c], all_initials[0], False)
        f_ev = self.scan_async(1, [addr, ci, ni], ~same, [eta, a, b, c], X0, False)
        (perm, t_perm), (ev, t_ev) = f_perm.result(), f_ev.result()
        self.ext_columns = [perm, ev]
        self.permutation_terminal = t_perm
        self.evaluation_terminal = t_ev
