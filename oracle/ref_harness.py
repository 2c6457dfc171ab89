"""oracle/ref_harness.py -- TEST INFRASTRUCTURE ONLY; runs ONLY in the build container.

Imports the upstream reference's own Python modules from /root/reference (read-only) so
that ``make_golden.py`` can record their outputs as fixtures.  Nothing from the reference
is copied into this repository and nothing here travels to the GPU box in a usable form:
the module refuses to load when /root/reference is absent.

Harness-side shims (the reference files stay untouched):
 * lib/tensorlist.py:169-176 ``TensorList.__getattr__`` answers for every attribute name
   torch.Tensor has, including ``__torch_function__``, which makes torch>=1.7 crash in
   ``torch.autograd.grad`` (model/optimizer.py:84).  We make it raise AttributeError for
   dunder names.
 * model/tracker.py imports cv2 and the NPP JIT extension through lib/image.py:5-6; both
   are absent here and are stubbed *before* the import.  torch.cuda.synchronize /
   empty_cache (tracker.py:65,126,159) are no-ops on this CPU-only box.
"""
import os
import sys
import types

REF_ROOT = '/root/reference'
if not os.path.isdir(REF_ROOT):
    raise ImportError('oracle/ref_harness.py needs the upstream reference at %s (build container only)' % REF_ROOT)

sys.dont_write_bytecode = True
if REF_ROOT not in sys.path:
    sys.path.insert(0, REF_ROOT)

import torch  # noqa: E402

from lib import tensorlist as _tl  # noqa: E402  (reference module)

_orig_getattr = _tl.TensorList.__getattr__


def _safe_getattr(self, name):
    if name.startswith('__') and name.endswith('__'):
        raise AttributeError(name)
    return _orig_getattr(self, name)


_tl.TensorList.__getattr__ = _safe_getattr

sys.modules.setdefault('cv2', types.ModuleType('cv2'))
_npp = types.ModuleType('lib._npp')
_npp.nppig_cpp = None
sys.modules.setdefault('lib._npp', _npp)
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None

from lib.tensorlist import TensorList  # noqa: E402,F401
from model.discriminator import Discriminator, DiscriminatorLoss  # noqa: E402,F401
from model.optimizer import GaussNewtonCG  # noqa: E402,F401
from model.memory import Memory  # noqa: E402,F401
from model.seg_network import SegNetwork  # noqa: E402,F401
from model.tracker import Tracker  # noqa: E402,F401


class AttrDict(dict):
    """easydict is absent; model/tracker.py:21-22 needs both ** and attribute access."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
