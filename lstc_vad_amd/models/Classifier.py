"""Drop-in for the reference ``models/Classifier.py`` (:5-23): the LTN head (2-way softmax)."""
from torch import nn

from ..functional import HeadFunction


class Classifier(nn.Module):
    def __init__(self, input_feature_dim, dropout_rate=0.6, weight_init=True):
        super().__init__()
        self.classifier = nn.Sequential(nn.Linear(input_feature_dim, 512), nn.ReLU(), nn.Dropout(dropout_rate),
                                        nn.Linear(512, 32), nn.Dropout(dropout_rate),
                                        nn.Linear(32, 2), nn.Softmax(dim=-1))
        if weight_init:
            for p in self.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)

    def forward(self, x):
        seq = self.classifier
        cfg = dict(dropout=seq[2].p, training=self.training, site="classifier")
        return HeadFunction.apply(x, seq[0].weight, seq[0].bias, seq[3].weight, seq[3].bias, seq[5].weight,
                                  seq[5].bias, cfg)
