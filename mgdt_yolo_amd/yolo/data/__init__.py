"""Device-side pieces of the reference's data path that sit directly in front of the model (reference: yolo/data/augment.py LetterBox)."""
from .augment import LetterBox  # noqa: F401
