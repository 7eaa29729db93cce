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
        rows, last = self._rows_and_last()
        pad = np.zeros((3, self._padding_length(rows)), dtype=np.uint64)
        pad[0] = last[0] if rows else 0                                # the last address repeats (instruction_table.py:19-25)
        self._pad_to(pad)

    def _scans(self, all_challenges, all_initials):
        """instruction_table.py:167-231"""
        a, b, c, d, e, f, alpha, beta, gamma, delta, eta = all_challenges
        product_rows, evaluation_rows = self._scan_masks()
        return [dict(kind=0, cols=[0, 1, 2], mask=product_rows, constants=[alpha, a, b, c], initial=all_initials[0], before=False),
                dict(kind=1, cols=[0, 1, 2], mask=evaluation_rows, constants=[eta, a, b, c], initial=X0, before=False)]

    def _make_scan_masks(self):
        m = self.base_array()
        addr, ci = m[0], m[1]
        same = np.concatenate([[False], addr[1:] == addr[:-1]]) if len(addr) else np.zeros(0, dtype=bool)
        # the running product absorbs a row when it is not padding and repeats the previous row's address (:197-205);
        # the running evaluation absorbs the first row of every address (:209-214); both are recorded AFTER the row's update
        return [(ci != 0) & same, ~same]

    def _after_extend(self, terminals, all_challenges, read):
        self.permutation_terminal, self.evaluation_terminal = terminals
