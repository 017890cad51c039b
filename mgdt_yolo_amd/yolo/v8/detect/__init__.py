from .predict import DetectionPredictor
from .val import DetectionValidator

__all__ = ('DetectionPredictor', 'DetectionValidator')
