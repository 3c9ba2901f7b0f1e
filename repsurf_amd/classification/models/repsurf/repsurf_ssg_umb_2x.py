"""RepSurf-U 2x: 4 abstraction stages, doubled widths, 6.8 M parameters —
`Model(args)` with the reference's interface (classification/models/repsurf/repsurf_ssg_umb_2x.py)."""
from models.repsurf._builder import UmbrellaClassifier

STAGES = [
    dict(npoint=512, radius=0.1, nsample=24, mlp=[128, 128, 256]),
    dict(npoint=128, radius=0.2, nsample=24, mlp=[256, 256, 512]),
    dict(npoint=32, radius=0.4, nsample=24, mlp=[512, 512, 1024]),
    dict(mlp=[1024, 1024, 2048]),
]


class Model(UmbrellaClassifier):
    def __init__(self, args):
        super().__init__(args, STAGES, head_in=2048)
