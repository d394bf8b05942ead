"""Host-side mirror of the reference's ``models`` package (same class names, constructor signatures,
attribute names and ``state_dict`` keys — SURVEY.md 8b) with every forward routed to liblstc_hip.so."""
from .Encoder import Encoder
from .EncoderLayer import EncoderLayer
from .MultiHeadAttention import MultiHeadAttention
from .FFN import PositionwiseFeedForward
from .Regressor import Regressor
from .Classifier import Classifier

__all__ = ["Encoder", "EncoderLayer", "MultiHeadAttention", "PositionwiseFeedForward", "Regressor", "Classifier"]
