"""Seeded synthetic inputs: the generators live in the package (findtextcenternet_amd/synth.py) so that bench.py does not
import from tests/; the tests keep importing ``synth``."""
from findtextcenternet_amd.synth import *  # noqa: F401,F403
from findtextcenternet_amd.synth import ADAMW_CASES, adamw_case, detector_maps, noise_images, page_images, page_uint8  # noqa: F401
