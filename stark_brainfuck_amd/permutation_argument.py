"""Permutation argument between two tables -- mirror of the reference's `permutation_argument.py`
(/root/reference/code/permutation_argument.py:4-34): the difference of two running-product columns must vanish at
the first row, so (lhs - rhs) / (x - 1) is a polynomial."""
import ctypes

from . import _lib
from .device import DeviceBuffer, current_stream


class PermutationArgument:
    def __init__(self, all_tables, lhs, rhs):
        self.all_tables = all_tables
        self.lhs = lhs
        self.rhs = rhs

    def quotient(self, fri_domain):
        """extension codeword (three limb planes) of the difference quotient, in HBM"""
        n = fri_domain.length
        out = DeviceBuffer(3 * n)
        lt, rt = self.all_tables[self.lhs[0]], self.all_tables[self.rhs[0]]
        _lib.check(_lib.load().bfs_difference_quotient(lt.ext_codeword_ptr(self.lhs[1]), rt.ext_codeword_ptr(self.rhs[1]), out.ptr,
                                                       n.bit_length() - 1, fri_domain.offset.value, fri_domain.omega.value, current_stream()))
        return out

    def combine_into(self, fri_domain, weight, accumulator, inv_x_minus_1=None, rows=None):
        """bfs_difference_combine: accumulator += (wa + wb x^shift) * (lhs - rhs) / (x - 1) without writing the quotient codeword;
        rows = (first, count): only at those points of the domain"""
        n = fri_domain.length
        first, count = (0, n) if rows is None else rows
        wa, wb, shift = weight
        w = _lib.CombWeight()
        w.wa, w.wb, w.shift = (ctypes.c_uint64 * 3)(*wa), (ctypes.c_uint64 * 3)(*wb), shift
        lt, rt = self.all_tables[self.lhs[0]], self.all_tables[self.rhs[0]]
        _lib.check(_lib.load().bfs_difference_combine_rows(lt.ext_codeword_ptr(self.lhs[1]), rt.ext_codeword_ptr(self.rhs[1]), n.bit_length() - 1,
                                                           fri_domain.offset.value, fri_domain.omega.value, ctypes.byref(w), accumulator.ptr,
                                                           inv_x_minus_1, first, count, current_stream()))

    def evaluate_difference(self, points):
        from .air import xsub
        return xsub(points[self.lhs[0]][self.lhs[1]], points[self.rhs[0]][self.rhs[1]])

    def quotient_degree_bound(self):
        return max(self.all_tables[self.lhs[0]].interpolant_degree(), self.all_tables[self.rhs[0]].interpolant_degree()) - 1
