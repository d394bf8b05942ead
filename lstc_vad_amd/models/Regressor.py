"""Drop-in for the reference ``models/Regressor.py`` (:4-21): the STN head."""
from torch import nn

from ..functional import HeadFunction


class Regressor(nn.Module):
    def __init__(self, input_feature_dim, dropout_rate=0.6, hidden_dim=512, weight_init=True):
        super().__init__()
        # nn.Sequential keeps the reference's state_dict keys regressor.{0,3,5}.{weight,bias}
        self.regressor = nn.Sequential(nn.Linear(input_feature_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout_rate),
                                       nn.Linear(hidden_dim, 32), nn.Dropout(dropout_rate),
                                       nn.Linear(32, 1), nn.Sigmoid())
        if weight_init:
            for p in self.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)

    def forward(self, x):
        seq = self.regressor
        cfg = dict(dropout=seq[2].p, training=self.training, site="regressor")
        return HeadFunction.apply(x, seq[0].weight, seq[0].bias, seq[3].weight, seq[3].bias, seq[5].weight,
                                  seq[5].bias, cfg)
