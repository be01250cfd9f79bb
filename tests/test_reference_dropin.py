"""B1 drop-in: the reference's OWN, unmodified objects-mode surgery
(`/root/reference/oadp/oake/objects.py:285-314`, `Validator._build_model`) runs against `oadp_amd.clip`
after nothing but the import swap — `import clip` resolving to `oadp_amd.clip`.

Build-container test in the style of tools/gen_golden.py: the reference is imported from where it lies
under import stubs for its un-vendored dependencies (todd, torchvision); nothing of it is copied or
shipped, and the test skips where /root/reference does not exist (the GPU box).  The checkpoint the
reference's `clip.load_default(False)` call picks up is a tiny synthetic TorchScript archive pointed to
by OAKE_CLIP_CHECKPOINT — the same ingestion path as `pretrained/clip/ViT-B-32.pt` (README.md:129).
"""
import importlib.util
import pathlib
import sys

import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
REF = pathlib.Path('/root/reference')

pytestmark = pytest.mark.skipif(not (REF / 'oadp' / 'oake' / 'objects.py').exists(),
                                reason='needs the reference checkout (build container only)')

TINY = dict(image_size=224, patch_size=32, width=128, layers=3, heads=2, mlp_dim=512, embed_dim=64)


@pytest.fixture
def reference_objects(tmp_path, monkeypatch):
    """The reference's oadp.oake.objects module with `clip` = oadp_amd.clip; sys.modules restored after."""
    from tests import _synth
    before = dict(sys.modules)
    spec = importlib.util.spec_from_file_location('_gen_golden', ROOT / 'tools' / 'gen_golden.py')
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    gg.install_stubs()  # todd / torchvision / package shells (and a clip stub, replaced next)
    import oadp_amd.clip
    import oadp_amd.clip.model
    sys.modules['clip'] = oadp_amd.clip            # <- the import swap INTEGRATION.md §A describes
    sys.modules['clip.model'] = oadp_amd.clip.model
    ckpt = tmp_path / 'ViT-tiny.pt'
    _synth.save_torchscript_checkpoint(_synth.tiny_state_dict(**TINY), ckpt)
    monkeypatch.setenv('OAKE_CLIP_CHECKPOINT', str(ckpt))
    monkeypatch.delenv('OAKE_SYNTHETIC_WEIGHTS', raising=False)
    try:
        gg.load_ref('base')
        yield gg.load_ref('objects')
    finally:
        for k in list(sys.modules):
            if k not in before:
                del sys.modules[k]
        sys.modules.update(before)


def test_reference_build_model_runs_unmodified(reference_objects):
    model, preprocess = reference_objects.Validator._build_model()
    from oadp_amd.clip.model import CLIP, VisionTransformer
    assert isinstance(model, CLIP) and isinstance(model.visual, VisionTransformer)
    v = model.visual
    # geometry surgery (objects.py:292-301)
    assert v.grid == 14
    assert tuple(v.conv1.stride) == (16, 16) and tuple(v.conv1.padding) == (15, 15)
    assert isinstance(v.positional_embedding, torch.nn.Parameter)
    assert tuple(v.positional_embedding.shape) == (197, TINY['width'])
    # the five registrations (objects.py:303-312) were recorded on the facade ...
    t = v.transformer
    assert [h.__name__ for h in v._forward_pre_hooks] == ['visual_forward_pre']
    assert [h.__name__ for h in t._forward_pre_hooks] == ['transformer_forward_pre']
    assert [h.__name__ for h in t._forward_hooks] == ['transformer_forward']
    assert len(t.resblocks) == TINY['layers']
    for blk in t.resblocks:
        assert [h.__name__ for h in blk._forward_pre_hooks] == ['residual_attention_block_forward_pre']
        assert blk.attn.num_heads == TINY['heads']
    owner = v._forward_pre_hooks[0].__self__
    assert type(owner).__name__ == 'Hooks' and type(owner).__module__ == 'oadp.oake.objects'
    # ... and recognised as the object-token stream the library implements (oake_encode_objects)
    assert v._hook_mode() == 'objects' and v._objects_mode()
    assert model.dtype == torch.float16
    with pytest.raises(ValueError):  # an objects-mode model refuses the plain call, as a shape error would upstream
        model.encode_image(torch.zeros(1, 3, 224, 224))
    assert preprocess is not None


def test_unknown_hook_patterns_are_refused(reference_objects):
    """Only the reference's pattern maps onto the native encoder; anything else fails loudly at forward
    time instead of being silently ignored."""
    import oadp_amd.clip as clip
    model, _ = clip.load_default(False)
    v = model.visual
    assert v._hook_mode() == 'none' and not v._objects_mode()
    handle = v.transformer.resblocks[0].register_forward_pre_hook(lambda m, i: None)
    with pytest.raises(NotImplementedError):
        v(torch.zeros(1, 3, 224, 224))
    handle.remove()
    assert v._hook_mode() == 'none'
    # a partial registration of the reference's own hooks is not the pattern either
    hooks = reference_objects.Hooks()
    h1 = v.register_forward_pre_hook(hooks.visual_forward_pre)
    with pytest.raises(NotImplementedError):
        v._hook_mode()
    h1.remove()
    assert v._hook_mode() == 'none'
