"""Instruction table -- mirror of the reference's `instruction_table.py` (/root/reference/code/instruction_table.py):
padding (:19-25) and `extend` (:167-231).  Constraints: air.InstructionAir."""
from . import air
from .air import xadd, xmul, xsub, xscale, xlift, X0, X1, xneg
from .table import Table


class InstructionTable(Table):
    address, current_instruction, next_instruction, permutation, evaluation = range(5)
    air = air.TABLE_AIRS[1]
    table_index = 1

    def __init__(self, field, length, num_randomizers, generator, order):
        super().__init__(field, 3, 5, length, num_randomizers, generator, order)

    def pad(self):
        rows = [list(r) for r in self.base_rows()]
        while len(rows) & (len(rows) - 1):
            rows.append([rows[-1][0], 0, 0])
        self._append_rows(rows)

    def extend(self, all_challenges, all_initials):
        a, b, c, d, e, f, alpha, beta, gamma, delta, eta = all_challenges
        perm = all_initials[0]
        ev = X0
        prev_addr = None
        rows, ext = self.base_rows(), []
        for i, (addr, ci, ni) in enumerate(rows):
            # the running product absorbs a row when it is not padding and repeats the previous row's address (:197-205)
            if ci != 0 and i > 0 and addr == rows[i - 1][0]:
                perm = xmul(perm, xsub(xsub(xsub(alpha, xscale(a, addr)), xscale(b, ci)), xscale(c, ni)))
            if prev_addr is None or addr != prev_addr:
                ev = xadd(xadd(xadd(xmul(eta, ev), xscale(a, addr)), xscale(b, ci)), xscale(c, ni))
            ext.append([perm, ev])
            prev_addr = addr
        self.ext_rows = ext
        self.permutation_terminal = perm
        self.evaluation_terminal = ev
