"""Encoder-layer harness: constructor / state-dict surface against the reference layer's own
state dicts (tests/golden/encoder_layer_small.npz, written by oracle/make_goldens_encoder.py).  No GPU."""
import ast
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "encoder_layer_small.npz")


def load_case(z, name):
    use_ffn, ffn_type, ffn_param, single_ffn, mode, n_vis = [str(x) for x in z[f"{name}/cfg"]]
    cfg = dict(use_ffn=use_ffn == "True", ffn_type=ffn_type, ffn_param=ast.literal_eval(ffn_param),
               single_ffn=ast.literal_eval(single_ffn))
    sd = {k[len(name) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{name}/sd/")}
    grads = {k[len(name) + 6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{name}/grad/")}
    return cfg, mode, int(n_vis), sd, grads


def build_layer(z, cfg, n_vis):
    from eve_amd.encoder import MultiwayTransformerEnocderLayer
    return MultiwayTransformerEnocderLayer(int(z["M"]), "quick_gelu", int(z["heads"]), int(z["I"]),
                                           (max(n_vis, 1), int(z["n_text"])), 0.0, layer_scale_init_values=0.1, **cfg)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_state_dict_keys_and_shapes_are_the_reference_layers(gold):
    for name in [str(c) for c in gold["cases"]]:
        cfg, mode, n_vis, sd, _ = load_case(gold, name)
        lay = build_layer(gold, cfg, n_vis)
        mine = lay.state_dict()
        assert set(mine) == set(sd), (name, set(mine) ^ set(sd))
        for k, v in sd.items():
            assert tuple(mine[k].shape) == tuple(v.shape), (name, k)
        lay.load_state_dict(sd, strict=True)
        # parameters in the same order as the reference's named_parameters() (optimizer param groups rely on names)
        assert [k for k, _ in lay.named_parameters()] == [k for k in sd if k in dict(lay.named_parameters())]


def test_eve_base_encoder_has_the_config_layout():
    from eve_amd.encoder import eve_base_encoder
    enc = eve_base_encoder()
    assert len(enc.layers) == 12
    moe_layers = [i for i, l in enumerate(enc.layers) if hasattr(l, "mlp_moe")]
    assert moe_layers == [10, 11]
    for i, l in enumerate(enc.layers):
        if i in moe_layers:
            assert len(l.mlp_moe.moe.experts.experts) == 32 and hasattr(l, "norm2_vl") and not hasattr(l, "mlp_imag")
        else:
            assert all(hasattr(l, n) for n in ("mlp_imag", "mlp_text", "norm2_imag", "norm2_text"))
    keys = enc.state_dict().keys()
    assert "layers.10.mlp_moe.moe.experts.experts.31.fc2.weight" in keys and "layers.0.mlp_text.fc1.bias" in keys
    assert not any("_split" in k for k in keys)
    # drop-path rates rise linearly to 0.1 (reference :429)
    assert enc.layers[0].drop_path.__class__.__name__ == "Identity" and abs(enc.layers[11].drop_path.drop_prob - 0.1) < 1e-6
    # fix_init_weight reaches the packed expert storage through the per-expert Parameters
    w = enc.layers[11].mlp_moe.moe.experts.experts[3].fc2.weight
    before = w.detach().clone()
    enc.fix_init_weight()
    torch.testing.assert_close(w.detach(), before / (2.0 * 12) ** 0.5)


def test_gate_loss_aggregation_matches_the_task_models_formula():
    from eve_amd.encoder import eve_base_encoder
    enc = eve_base_encoder()
    enc.layers[10].aux_loss_vl, enc.layers[10].aux_loss_vl_i = torch.tensor(3.0), 2
    enc.layers[11].aux_loss_vl, enc.layers[11].aux_loss_vl_i = torch.tensor(5.0), 1
    got = enc.gate_loss(0.001)
    assert abs(float(got) - 0.001 * (1.5 + 5.0) / 2) < 1e-9
    assert enc.layers[10].aux_loss_vl_i == 0 and enc.gate_loss(0.001) == 0.0


def test_deepcopy_keeps_the_split_helper_on_the_copied_modules():
    import copy
    from eve_amd.encoder import eve_base_encoder
    enc = eve_base_encoder()
    c = copy.deepcopy(enc.layers[0])
    assert c._split.mlp_imag is c.mlp_imag and c._split.norm2_text is c.norm2_text and c.mlp_imag is not enc.layers[0].mlp_imag
