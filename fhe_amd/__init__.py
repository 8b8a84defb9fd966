"""fhe_amd -- host side of liblfhe, the MI355X (gfx950) TFHE backend.

Only what the bootstrap path and the integer operators need: the HIP kernels and C ABI under
csrc/, their ctypes binding (native), the client that mirrors the reference SDK's method names
(client) and the HTTP surface of the reference engine (server)."""
from . import native  # noqa: F401
from .native import Context, LfheError, Params  # noqa: F401
