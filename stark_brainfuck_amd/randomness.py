"""Where the prover draws its random bytes from.

The reference calls `os.urandom` directly (brainfuck_stark.py:200-211, table.py:119-125, salted_merkle.py:12) and its tests replace the
module-level name to replay a byte stream; the modules here keep that name (`brainfuck_stark.urandom`, `table.urandom`,
`salted_merkle.urandom`) and draw through `source(urandom)`:

* normally that is the module's `urandom` -- the operating system's, or whatever a test assigned to the module attribute;
* inside `with override(stream):` it is `stream`, for the CURRENT context only (a `contextvars.ContextVar`: another prover thread of the
  same process, which include/bfstark.h supports, keeps its own source).  `shard.shared_randomness` uses this so that the ranks of a
  cooperative proof read one broadcast stream without touching process-global state.
"""
import contextvars

_override = contextvars.ContextVar("bfs_randomness_override", default=None)


def source(module_urandom):
    """the callable count -> bytes this context draws from"""
    stream = _override.get()
    return module_urandom if stream is None else stream


class override:
    """context manager: `stream` (count -> bytes) is the random source of this context until exit"""

    def __init__(self, stream):
        self.stream, self._token = stream, None

    def __enter__(self):
        self._token = _override.set(self.stream)
        return self.stream

    def __exit__(self, *exc):
        _override.reset(self._token)
        return False
