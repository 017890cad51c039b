"""reference: yolo/v8/detect/predict.py - the detection task's predictor plug-in."""
from ...engine.predictor import DetectionPredictor

__all__ = ('DetectionPredictor',)
