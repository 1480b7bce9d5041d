"""inr_mi355x -- MI355X-native engine for the coordinate-MLP fitting hot path of
luisdavid64/MRI-Implicit-Neural-Representations: drop-in mirrors of SIREN / FFN / WIRE / WIRE2D (networks.py),
FourierNet / GaborNet / KGaborNet / MultiscaleKFourier / MultiscaleBoundedFourier (mfn.py), the two training loops
(train.py, train_kspace_multiscale.py), the ring ensemble, data ingest and the evaluation chain -- see DESIGN.md for
the scope table.  The arithmetic lives in lib/libinr_mi355x.so (hand-written gfx950 HIP kernels behind the C-ABI
of include/inr_abi.h); this package is the thin host side and has no CPU path."""
from .networks import SIREN, FFN, WIRE, WIRE2D, Positional_Encoder  # noqa: F401
from .engine import MLPEngine, LossSpec, encode_gauss  # noqa: F401
