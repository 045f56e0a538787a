"""buffalo.algo-compatible front: ALS / BPRMF / WARP / CFR / EALS with the reference's option objects and
training loops, running on the MI355X backend (buffalo_amd.backend)."""
from .als import ALS  # noqa: F401
from .bpr import BPRMF  # noqa: F401
from .cfr import CFR  # noqa: F401
from .eals import EALS  # noqa: F401
from .options import ALSOption, BPRMFOption, CFROption, EALSOption, WARPOption  # noqa: F401
from .warp import WARP  # noqa: F401
