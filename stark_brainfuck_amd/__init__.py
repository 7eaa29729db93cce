"""stark_brainfuck_amd -- MI355X (gfx950) backend for the polynomial hot path of aszepieniec/stark-brainfuck.

Same call surface as the reference's flat modules:

    from stark_brainfuck_amd import (BaseField, BaseFieldElement, ExtensionField, ExtensionFieldElement, Polynomial,
                                     ntt, intt, fast_multiply, fast_coset_evaluate, fast_coset_interpolate,
                                     batch_inverse, fast_coset_divide, Merkle, SaltedMerkle, ProofStream, Fri)

All bulk work runs in hand-written HIP kernels behind the C ABI of libbfstark_hip.so (include/bfstark.h); there is
no CPU fallback -- importing the compute entry points without the built library raises BackendUnavailable.
"""
from .algebra import BaseField, BaseFieldElement, xgcd
from .univariate import Polynomial, colinear        # (the reference's name for it, `test_colinearity`, stays in .univariate only: a star-import
#                                                       of it into a test module makes pytest collect it -- SURVEY.md 4)
from .extension_field import ExtensionField, ExtensionFieldElement
from .arrays import BaseArray, XArray
from .ntt import (ntt, intt, fast_multiply, fast_coset_evaluate, fast_coset_interpolate, batch_inverse,
                  fast_coset_divide, fast_zerofier, fast_evaluate, fast_interpolate)
from .merkle import Merkle
from .salted_merkle import SaltedMerkle
from .ip import ProofStream, reference_pickle
from .fri import Fri
from ._lib import BackendUnavailable

__all__ = ["BaseField", "BaseFieldElement", "xgcd", "Polynomial", "colinear", "ExtensionField",
           "ExtensionFieldElement", "BaseArray", "XArray", "ntt", "intt", "fast_multiply", "fast_coset_evaluate",
           "fast_coset_interpolate", "batch_inverse", "fast_coset_divide", "Merkle", "SaltedMerkle", "ProofStream",
           "reference_pickle", "Fri", "BackendUnavailable"]
