"""CPU: the C-ABI library builds, loads and exports what include/master_hip.h declares; host logic
(window tables, state-dict compatibility, error behaviour) matches the reference's contract."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import master_oracle as O

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "master_hip.h")).read()
    declared = set(re.findall(r"\b(master_[a-z0-9_]+)\s*\(", header))
    declared -= {"master_stream_t"}
    assert len(declared) >= 13
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in master_hip.h but not exported"
    assert lib.master_version() >= 100


def test_invalid_arguments_return_status_not_crash(lib):
    # argument validation happens before any HIP call, so this is safe without a GPU
    rc = lib.master_linear_fwd(None, None, None, None, None, None, 4, 4, 32, 32, 4, 0, 0, 0, None)
    assert rc == -1 and b"null" in lib.master_last_error_string()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.master_linear_fwd(p, p, None, None, None, p, 4, 4, 33, 33, 4, 0, 0, 0, None)
    assert rc == -1 and b"K%32" in lib.master_last_error_string()
    rc = lib.master_patch_embed_fwd(p, p, p, p, p, p, 1, 8, 8, 96, 1e-5, 0, None)
    assert rc == -2      # MASTER_ERR_UNSUPPORTED


def test_missing_library_fails_loudly(monkeypatch):
    from mastermetastyletransfer_amd import ops
    monkeypatch.setattr(ops, "_lib", None)
    monkeypatch.setattr(ops, "LIB_PATH", "/nonexistent/libmaster_hip.so")
    with pytest.raises(ops.MasterHipError, match="no CPU fallback"):
        ops.load_library()


@pytest.mark.parametrize("H,W,shift", [(7, 7, (0, 0)), (14, 14, (3, 3)), (9, 9, (4, 4)), (9, 12, (4, 4)), (14, 9, (4, 0)),
                                       (32, 32, (4, 4)), (64, 64, (3, 3)), (64, 64, (0, 0)), (5, 20, (4, 4))])
def test_window_tables_match_oracle(H, W, shift):
    from mastermetastyletransfer_amd.window_plan import window_plan
    plan = window_plan(H, W, (7, 7), shift, "cpu")
    src, region, pH, pW = O.window_plan(H, W, (7, 7), tuple(shift))
    want = torch.where(src == H * W, torch.full_like(src, -1), src)
    assert plan.nW == src.shape[0] and plan.N == 49 and plan.n_pad == pH * pW - H * W
    assert torch.equal(plan.src[:, :49].long(), want) and (plan.src[:, 49:] == -1).all()
    if region is None:
        assert plan.region is None
    else:
        # region ids only matter up to equality within a window
        a, b = plan.region[:, :49].long(), region
        assert torch.equal(a[:, :, None] == a[:, None, :], b[:, :, None] == b[:, None, :])


def test_relative_position_index_matches_reference_formula():
    from mastermetastyletransfer_amd.window_plan import relative_position_index
    assert torch.equal(relative_position_index((7, 7)), O.relative_position_index((7, 7)))
    assert torch.equal(relative_position_index((4, 6)), O.relative_position_index((4, 6)))


def test_state_dict_keys_and_checkpoint_roundtrip(tmp_path, lib):
    import warnings
    from mastermetastyletransfer_amd import MasterStyleTransferModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7])
    for sd, shapes in ((m.style_transformer.state_dict(), O.style_transformer_shapes()),
                       (m.decoder.state_dict(), O.decoder_shapes()), (m.swin_encoder.state_dict(), O.swin_shapes())):
        assert set(sd) == set(shapes)
        assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert sum(p.numel() for p in m.style_transformer.parameters()) == 3422424
    assert sum(p.numel() for p in m.decoder.parameters()) == 877091
    # checkpoints written the way the reference's trainer writes them load through the same ctor kwargs
    torch.save(m.style_transformer.state_dict(), tmp_path / "st.pt")
    torch.save(m.decoder.state_dict(), tmp_path / "dec.pt")
    torch.save(m.swin_encoder.state_dict(), tmp_path / "swin.pt")
    m2 = MasterStyleTransferModel(project_absolute_path=str(tmp_path), swin_model_relative_path="swin.pt",
                                  style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7],
                                  direct_pretrained_style_transformer_path=str(tmp_path / "st.pt"),
                                  direct_pretrained_decoder_path=str(tmp_path / "dec.pt"))
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)


def test_error_behaviour_mirrors_reference(lib):
    import warnings
    from mastermetastyletransfer_amd import Decoder, MasterStyleTransferModel, ShiftedWindowAttention
    with pytest.raises(ValueError, match="length 2"):
        ShiftedWindowAttention(256, 8, [7], [4, 4])
    with pytest.raises(AssertionError, match="Invalid initializer"):
        Decoder(256, "nope")
    with pytest.raises(ValueError, match="Invalid Swin Transformer variant"):
        MasterStyleTransferModel(swin_variant="swin_X")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(ValueError, match="path of the pretrained weights"):
            MasterStyleTransferModel(style_transformer_load_pretrained_weights=True)
        with pytest.raises(NotImplementedError):
            MasterStyleTransferModel(style_decoder_use_regular_MHA_instead_of_Swin_at_the_end=True)
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7]).eval()
    from mastermetastyletransfer_amd.ops import MasterHipError
    with pytest.raises(MasterHipError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))
    m.train()
    with pytest.raises(NotImplementedError, match="train-mode"):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mastermetastyletransfer_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, re.M), fn
