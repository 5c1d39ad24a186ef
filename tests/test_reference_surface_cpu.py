"""INTEGRATION.md section 1, executed as far as a box without a GPU can: the REFERENCE's own model classes
(/root/reference/src/models: DMCI, the LD DMC, the HT DMC) run against this repo's two plug-in modules.

  * `MLCodec_extensions_cpp` (dcvc_amd/plugin): the reference's CompressionModel.update() builds its entropy tables
    through it (entropy_models.py:34-43) - executed for real (host code), tables compared with the ones this
    repo's operator mirror builds;
  * `inference_extensions_cuda` (dcvc_amd/plugin): the reference's compress() / add_ref_feature_from_frame()
    import it by that name, construct the proxy class and call set_param(state_dict + CDF tensors, skip_thres)
    (image_model.py:194-206, video_model_ld.py:277-289, video_model_ht.py:413-430). The C ABI entry points behind
    create / set_param are replaced by recorders (they need a device), everything in front of them - the import,
    the class and method names, the argument marshalling of dcvc_amd/plugin - is the real code. What set_param
    receives must be exactly the parameter inventory the native codec parses (dcvc_amd/arch.py) plus the four
    CDF tensors as int32.

The reference tree exists only in the build container; on the GPU box this module is skipped (the -m gpu tests
drive the same plug-in classes through dcvc_amd/models.py, the mirror of these reference classes)."""
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "dcvc_amd", "plugin")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "models")),
                                reason="the reference tree is not on this box")


@pytest.fixture(scope="module")
def ref_models():
    added = [p for p in (REF, PLUGIN) if p not in sys.path]
    for p in added:
        sys.path.insert(0, p)
    try:
        from src.models.image_model import DMCI
        from src.models.video_model_ht import DMC as DMCHT
        from src.models.video_model_ld import DMC as DMCLD
        yield {"dmci": DMCI, "ld": DMCLD, "ht": DMCHT}
    finally:
        for p in added:
            sys.path.remove(p)
        for name in [n for n in sys.modules if n == "src" or n.startswith("src.")]:
            del sys.modules[name]


class _Recorder:
    def __init__(self):
        self.params, self.skip_thres, self.created = None, None, 0

    def create(self, *a):
        self.created += 1
        return 0x1234

    def set_param(self, handle, n, names, ptrs, dtypes, ndims, dims, skip_thres):
        assert handle == 0x1234
        shapes, k = [], 0
        for i in range(n):
            shapes.append(tuple(int(dims[k + j]) for j in range(ndims[i])))
            k += ndims[i]
        self.params = {names[i].decode(): (int(dtypes[i]), shapes[i]) for i in range(n)}
        self.skip_thres = float(skip_thres)
        return 0


def _patched(monkeypatch, table):
    import inference_extensions_cuda as plug
    rec = _Recorder()
    monkeypatch.setitem(table(plug), "create", rec.create)
    monkeypatch.setitem(table(plug), "set_param", rec.set_param)
    monkeypatch.setitem(table(plug), "destroy", lambda h: None)
    return plug, rec


def _check_inventory(rec, spec, model):
    from inference_extensions_cuda import _DTYPES
    cdf = {"gaussian_encoder.quantized_cdf", "gaussian_encoder.cdf_length", "bit_estimator_z.quantized_cdf", "bit_estimator_z.cdf_length"}
    got = set(rec.params)
    assert cdf <= got, "the four CDF tensors travel inside the state dict (common_model.py:64-70)"
    weights = got - cdf
    assert weights == set(spec), (sorted(weights - set(spec))[:5], sorted(set(spec) - weights)[:5])
    for name, shape in spec.items():
        assert rec.params[name][1] == tuple(shape), name
    for name in cdf:
        assert rec.params[name][0] == _DTYPES[torch.int32], name
    assert rec.skip_thres == pytest.approx(model.gaussian_encoder.skip_thres)


def test_reference_dmci_drives_both_plugins(ref_models, monkeypatch):
    from dcvc_amd import arch, models
    net = ref_models["dmci"]().eval()
    net.update(0.15)                                     # the reference's update() through OUR MLCodec_extensions_cpp
    mirror = models.DMCI()
    mirror.load_state_dict(net.state_dict())
    mirror.update(0.15)
    for a, b in zip(net.gaussian_encoder.get_cdf_info() + net.bit_estimator_z.get_cdf_info(),
                    [np.asarray(t) for t in mirror.get_cdf_info()][2:] + [np.asarray(t) for t in mirror.get_cdf_info()][:2]):
        assert np.array_equal(np.asarray(a), b), "entropy tables: reference update() vs the operator mirror"
    plug, rec = _patched(monkeypatch, lambda p: p._F)
    x = torch.zeros(1, 3, 64, 64).half()
    with pytest.raises(ValueError, match="CUDA fp16"):    # the first thing that needs a device: the picture itself
        net.half().compress(x, 32, 0, 0)
    assert rec.created == 1 and isinstance(net.proxy, plug.DMCIProxy)
    _check_inventory(rec, arch.dmci_spec(), net)


@pytest.mark.parametrize("kind", ["ld", "hts", "htl"])
def test_reference_inter_models_reach_set_param(ref_models, monkeypatch, kind):
    from dcvc_amd import arch
    if kind == "ld":
        net, spec, table, cls = ref_models["ld"]().eval(), arch.dmc_ld_spec(), (lambda p: p._LD), "DMCLDProxy"
    else:
        from src.utils.common import ModelStructure        # the enum the reference harness passes (test_video.py:442-447)
        net, spec, table = ref_models["ht"](ModelStructure(kind)).eval(), arch.dmc_ht_spec(kind == "hts"), (lambda p: p._HT)
        cls = "DMCHTSProxy" if kind == "hts" else "DMCHTLProxy"
    net.update(0.15)
    plug, rec = _patched(monkeypatch, table)
    frame = torch.zeros(1, 3, 64, 64).half()
    with pytest.raises(ValueError, match="CUDA fp16"):
        net.half().add_ref_feature_from_frame(frame, True)
    assert rec.created == 1 and type(net.proxy).__name__ == cls
    _check_inventory(rec, spec, net)


def test_without_a_device_the_reference_sees_a_clean_error_not_a_missing_module(ref_models):
    """image_model.py:196-202 turns an ImportError into NotImplementedError('cannot import cuda implementation');
    with the plug-in on the path the import succeeds, and a box without a GPU fails where it should: creating
    the codec."""
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    from dcvc_amd import _lib
    net = ref_models["dmci"]().eval()
    net.update(0.15)
    with pytest.raises(_lib.DcvcError, match="no ROCm-capable device|hip"):
        net.half().compress(torch.zeros(1, 3, 64, 64).half(), 32, 0, 0)
