"""RepSurf-U (umbrella) on PointNet++ SSG, 3 abstraction stages, 1.48 M parameters —
`Model(args)` with the reference's interface (classification/models/repsurf/repsurf_ssg_umb.py)."""
from models.repsurf._builder import UmbrellaClassifier

STAGES = [
    dict(npoint=512, radius=0.2, nsample=32, mlp=[64, 64, 128]),
    dict(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 256]),
    dict(mlp=[256, 512, 1024]),
]


class Model(UmbrellaClassifier):
    def __init__(self, args):
        super().__init__(args, STAGES, head_in=1024)
