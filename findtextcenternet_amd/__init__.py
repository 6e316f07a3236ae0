"""findtextcenternet_amd -- MI355X (gfx950) native detector hot path of findtextCenterNet.

Public names mirror the reference (``/root/reference/models/detector.py``, ``util_func.py:5-9``)."""
from .schema import feature_dim, height, modulo_list, scale, width          # noqa: F401
from .detector import CenterNetDetection, CenterNetDetector, SimpleDecoder, TextDetectorModel   # noqa: F401
from .decode import Decoded, HipDetectorBackend, TileGeom, decode_peaks, exact_logit_cut, tile_keep_rect, tiles_to_device   # noqa: F401
from .weights import deterministic_state_dict, load_tf_efficientnetv2_npz  # noqa: F401
from .page import PageDetector, linedetect_parse, linedetect_request, page_merge_gpu   # noqa: F401
from .optim import AdamWScheduleFree   # noqa: F401
from .train_step import TrainStep   # noqa: F401
from .lanes import DetectorLanes   # noqa: F401
from . import synth   # noqa: F401

__all__ = ["TextDetectorModel", "CenterNetDetection", "CenterNetDetector", "SimpleDecoder", "HipDetectorBackend",
           "TileGeom", "Decoded", "decode_peaks", "tiles_to_device", "exact_logit_cut", "tile_keep_rect", "deterministic_state_dict", "load_tf_efficientnetv2_npz", "PageDetector", "page_merge_gpu", "linedetect_request",
           "linedetect_parse", "AdamWScheduleFree", "TrainStep", "DetectorLanes",
           "width", "height", "scale", "feature_dim", "modulo_list"]
