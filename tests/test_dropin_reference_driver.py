"""The upstream driver really drops onto the package (north_star: "the reference evaluate.py drops onto it unchanged").

BUILD CONTAINER ONLY: the upstream ``evaluate.py`` is imported from /root/reference where it lies (never copied; the test is skipped when
the reference is absent, e.g. on the GPU box).  INTEGRATION.md section A's aliasing is applied -- the package's module tree under the
upstream names ``model`` / ``lib`` -- plus a stand-in for ``easydict`` (absent in this image), the upstream file is executed, and its own
``Parameters`` / ``get_model()`` are driven: every class it imported must be the package's, its hyper-parameters must arrive in the package's
``Discriminator``, and ``load_state_dict`` must accept exactly the upstream ``refiner.*`` checkpoint keys (taken from the UPSTREAM
``SegNetwork``'s own state dict).  No GPU here: the extractor's upload to the device is stubbed, nothing is computed.
"""
import importlib
import importlib.util
import os
import sys
import types

import pytest
import torch

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'evaluate.py')), reason='upstream reference not present (build container only)')

PKG_MODEL = ('tracker', 'discriminator', 'optimizer', 'memory', 'feature_extractor', 'seg_network', 'augmenter')
PKG_LIB = ('tensorlist', 'utils', 'image', 'datasets', 'evaluation')


class _EasyDict(dict):
    """Stand-in for easydict.EasyDict (absent here): attribute access, nested dicts converted."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _EasyDict(v) if isinstance(v, dict) and not isinstance(v, _EasyDict) else v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


@pytest.fixture
def upstream_evaluate(monkeypatch):
    """INTEGRATION.md section A, then the upstream file executed as a module."""
    import frtm_vos_amd  # noqa: F401
    saved = {k: v for k, v in sys.modules.items() if k in ('model', 'lib', 'easydict') or k.startswith(('model.', 'lib.'))}
    for k in saved:
        del sys.modules[k]
    ed = types.ModuleType('easydict')
    ed.EasyDict = _EasyDict
    sys.modules['easydict'] = ed
    sys.modules['model'] = importlib.import_module('frtm_vos_amd.model')
    sys.modules['lib'] = importlib.import_module('frtm_vos_amd.lib')
    for m in PKG_MODEL:
        sys.modules['model.' + m] = importlib.import_module('frtm_vos_amd.model.' + m)
    for m in PKG_LIB:
        sys.modules['lib.' + m] = importlib.import_module('frtm_vos_amd.lib.' + m)
    spec = importlib.util.spec_from_file_location('upstream_evaluate', os.path.join(REF, 'evaluate.py'))
    mod = importlib.util.module_from_spec(spec)
    old_flag, sys.dont_write_bytecode = sys.dont_write_bytecode, True
    path0 = list(sys.path)
    try:
        spec.loader.exec_module(mod)             # runs the imports and the class definition; the __main__ block does not run
        yield mod
    finally:
        sys.dont_write_bytecode = old_flag
        sys.path[:] = path0
        for k in [k for k in sys.modules if k in ('model', 'lib', 'easydict') or k.startswith(('model.', 'lib.'))]:
            del sys.modules[k]
        sys.modules.update(saved)


def _upstream_refiner_state(in_channels):
    """State dict of the UPSTREAM SegNetwork (model/seg_network.py, pure torch: importable here), as 'refiner.*' checkpoint keys."""
    spec = importlib.util.spec_from_file_location('upstream_seg_network', os.path.join(REF, 'model', 'seg_network.py'))
    mod = importlib.util.module_from_spec(spec)
    lib = types.ModuleType('lib')
    lib_utils = types.ModuleType('lib.utils')
    ref_utils = importlib.util.spec_from_file_location('upstream_lib_utils', os.path.join(REF, 'lib', 'utils.py'))
    um = importlib.util.module_from_spec(ref_utils)
    hold = {k: sys.modules.get(k) for k in ('lib', 'lib.utils')}
    try:
        ref_utils.loader.exec_module(um)
        lib_utils.__dict__.update({k: v for k, v in um.__dict__.items() if not k.startswith('__')})
        lib.utils = lib_utils
        sys.modules['lib'], sys.modules['lib.utils'] = lib, lib_utils
        spec.loader.exec_module(mod)
    finally:
        for k, v in hold.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    from collections import OrderedDict
    chans = OrderedDict(layer5=in_channels * 2, layer4=in_channels, layer3=in_channels // 2, layer2=in_channels // 4)
    torch.manual_seed(5)
    net = mod.SegNetwork(1, 64, chans, True)
    return {'refiner.' + k: v.clone() for k, v in net.state_dict().items()}


def test_upstream_evaluate_drops_onto_the_package(upstream_evaluate, monkeypatch):
    ev = upstream_evaluate
    from frtm_vos_amd.model import augmenter, discriminator, feature_extractor, seg_network, tracker
    from frtm_vos_amd.lib import datasets, evaluation
    # (1) everything the upstream driver imported (evaluate.py:18-23) is the package's
    assert ev.Tracker is tracker.Tracker
    assert ev.ResnetFeatureExtractor is feature_extractor.ResnetFeatureExtractor
    assert ev.SegNetwork is seg_network.SegNetwork
    assert ev.ImageAugmenter is augmenter.ImageAugmenter
    assert ev.DAVISDataset is datasets.DAVISDataset and ev.YouTubeVOSDataset is datasets.YouTubeVOSDataset
    assert ev.evaluate_dataset is evaluation.evaluate_dataset
    assert ev.Parameters.__module__ == 'upstream_evaluate'                 # ... while Parameters is upstream's own code
    # (2) upstream Parameters on an upstream-shaped checkpoint (ResNet-18 variant: autodetected from the refiner's layer4 reduce conv, :38-44)
    weights = _upstream_refiner_state(256)
    assert weights['refiner.TSE.layer4.reduce.0.weight'].shape[1] == 256
    p = ev.Parameters(weights, device='cpu')
    assert p.feature_extractor == 'resnet18' and p.init_iters == (5, 10, 10, 10, 10) and p.update_iters == (10,)
    assert ev.Parameters(weights, fast=True, device='cpu').init_iters == (5, 10, 10, 10)
    # (3) get_model() (evaluate.py:91-105) builds the package's Tracker; no GPU here: the extractor's device upload is stubbed
    if not torch.cuda.is_available():
        monkeypatch.setattr(feature_extractor.ResnetFeatureExtractor, 'to', lambda self, device: self)
    else:
        p = ev.Parameters(weights, device='cuda:0')
    mdl = p.get_model()
    assert type(mdl) is tracker.Tracker and type(mdl.refiner) is seg_network.SegNetwork
    assert type(mdl.feature_extractor) is feature_extractor.ResnetFeatureExtractor and mdl.feature_extractor.name == 'resnet18'
    assert p.disc_params.in_channels == 256                                    # read back from get_out_channels() (:95)
    for k, v in weights.items():                                               # the checkpoint really arrived in the refiner
        assert torch.equal(mdl.state_dict()[k].cpu(), v), k
    assert set(mdl.state_dict().keys()) == set(weights.keys())                 # "exactly the refiner.* keys" 
    # (4) the upstream hyper-parameters arrive in the package's Discriminator through TargetObject(disc_params=...) (tracker.py:21-22)
    t = tracker.TargetObject(obj_id=1, disc_params=mdl.disc_params, index=1, start_frame=0, start_mask=None)
    d = t.discriminator
    assert type(d) is discriminator.Discriminator and t.disc_layer == 'layer4'
    assert d.project.weight.shape == (96, 256, 1, 1) and d.filter.weight.shape == (1, 96, 3, 3)
    assert d.init_iters == (5, 10, 10, 10, 10) and d.update_iters == (10,) and d.memory_size == 80 and d.train_skipping == 8
    assert d.filter_reg == (1e-4, 1e-2) and d.precond == (1e-4, 1e-2) and d.learning_rate == 0.1 and d.update_filters is True
    assert d.pw_params == dict(method='hinge', tf=0.1)
    assert abs(d.direction_forget_factor - 0.9 ** 750) < 1e-40
    # (5) strictness: a checkpoint with a missing or a foreign key is refused like upstream's nn.Module would
    bad = dict(weights)
    bad.pop('refiner.TSE.layer4.reduce.0.weight')
    with pytest.raises(RuntimeError):
        mdl.load_state_dict(bad)
    with pytest.raises(RuntimeError):
        mdl.load_state_dict(dict(weights, **{'feature_extractor.conv1.weight': torch.zeros(1)}))


def test_upstream_parameters_match_the_packages_own_driver(upstream_evaluate):
    """The package's evaluate.Parameters (its own driver) carries the same hyper-parameters as upstream's (evaluate.py:32-34,46-89)."""
    from frtm_vos_amd.evaluate import Parameters as Own
    w = {'refiner.TSE.layer4.reduce.0.weight': torch.zeros(64, 1024, 1, 1)}
    for fast in (False, True):
        up, own = upstream_evaluate.Parameters(w, fast=fast, device='cpu'), Own(w, fast=fast, device='cpu')
        assert up.feature_extractor == own.feature_extractor == 'resnet101'
        assert dict(up.disc_params) == {k: v for k, v in dict(own.disc_params).items() if k in up.disc_params}
        assert dict(up.refnet_params) == dict(own.refnet_params)

        def plain(d):
            return {k: (plain(v) if isinstance(v, dict) else v) for k, v in d.items()}
        assert plain(up.aug_params) == plain(own.aug_params)
