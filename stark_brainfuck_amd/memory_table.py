"""Memory table -- mirror of the reference's `memory_table.py` (/root/reference/code/memory_table.py): `derive_matrix`
(:20-38), padding (:40-44) and `extend` (:172-206).  Constraints: air.MemoryAir."""
import numpy as np

from . import air
from .table import Table, P, _val


class MemoryTable(Table):
    cycle, memory_pointer, memory_value, dummy, permutation = range(5)
    air = air.TABLE_AIRS[2]
    table_index = 2

    def __init__(self, field, length, num_randomizers, generator, order):
        super().__init__(field, 4, 5, length, num_randomizers, generator, order)

    @staticmethod
    def derive_matrix(processor_matrix):
        """rows (cycle, memory pointer, memory value, dummy) of every non-padding processor row, sorted by memory pointer
        (stable), with dummy rows inserted where the cycle count of one address jumps by more than one."""
        field = processor_matrix[0][0].field if hasattr(processor_matrix[0][0], "field") else None
        values = getattr(processor_matrix, "values", None)
        if values is not None:
            rows = [[r[0], r[4], r[5], 0] for r in values.tolist() if r[2] != 0]
        else:
            rows = [[_val(r[0]), _val(r[4]), _val(r[5]), 0] for r in processor_matrix if _val(r[2]) != 0]
        rows.sort(key=lambda r: r[1])
        # the reference inserts one dummy row at a time with list.insert (quadratic); same result in one pass: between two
        # rows of the same address whose cycle counts are not consecutive, dummy rows count the cycles up and keep the value
        out = []
        for k, row in enumerate(rows):
            out.append(row)
            if k + 1 < len(rows) and rows[k + 1][1] == row[1]:
                clk, target = row[0], rows[k + 1][0]
                while (clk + 1) % P != target:
                    clk = (clk + 1) % P
                    out.append([clk, row[1], row[2], 1])
        rows = out
        if field is None:
            return rows
        from .vm import _matrix
        return _matrix(rows, 4, field)

    def pad(self):
        rows, last = self._rows_and_last()
        k = self._padding_length(rows)
        pad = np.zeros((4, k), dtype=np.uint64)
        if k:
            pad[0] = self._counting(last[0], k)                         # dummy rows: cycle counts up, pointer and value stay (:40-44)
            pad[1], pad[2], pad[3] = last[1], last[2], 1
        self._pad_to(pad)

    def _scans(self, all_challenges, all_initials):
        """memory_table.py:172-206"""
        a, b, c, d, e, f, alpha, beta, gamma, delta, eta = all_challenges
        return [dict(kind=0, cols=[0, 1, 2], mask=self._scan_masks()[0], constants=[beta, d, e, f], initial=all_initials[1], before=True)]

    def _make_scan_masks(self):
        return [self.base_array()[3] == 0]                  # dummy rows (the clock jumped) leave the product alone

    def _after_extend(self, terminals, all_challenges, read):
        self.permutation_terminal = terminals[0]
