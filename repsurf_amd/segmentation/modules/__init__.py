"""Drop-in replacement of the reference's `segmentation/modules` package.

Put `repsurf_amd/segmentation` first on PYTHONPATH (the reference runs with PYTHONPATH=./ from its
`segmentation/` directory, scripts/s3dis/train_repsurf_umb.sh:3) and `from modules.repsurface_utils import ...`
in the reference's model files resolves here."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:          # make `import repsurf_amd` work when only this tree is on the path
    sys.path.insert(0, _ROOT)
