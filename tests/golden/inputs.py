"""Seeded input builders shared by make_golden.py (build container) and tests/test_oracle.py
(anywhere).  Only seeds are stored in the fixture; these functions regenerate the tensors."""
import torch

from oracle import cara_oracle as O


def seeded_backbone_into(model, seed):
    """Overwrite every backbone tensor from a seeded generator: NON-zero biases and NON-trivial
    LN affine so that every term of the path is exercised."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.startswith("CP_"):
                continue
            if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
                v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(0.02 * torch.randn(v.shape, generator=g))


def randomise_cp(cp, seed):
    """cp: dict name->tensor (or a module with CP_* attributes).  Makes the adapter non-zero."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    get = (lambda n: cp[n]) if isinstance(cp, dict) else (lambda n: getattr(cp, n))
    with torch.no_grad():
        get("CP_A2").copy_(0.05 * torch.randn(get("CP_A2").shape, generator=g))
        get("CP_P2").copy_(0.05 * torch.randn(get("CP_P2").shape, generator=g))
        for n in ("CP_bias1", "CP_bias2", "CP_bias3"):
            get(n).copy_(0.02 * torch.randn(get(n).shape, generator=g))


def oracle_case(global_seed, backbone_seed, cp_seed, rank, depth, img_size, l_mu=1.5, l_std=0.1,
                num_classes=100, drop_path_rate=0.1):
    """Rebuild (weights, cp) exactly as make_golden.py built them around the reference's
    ``cara()`` call: same global-RNG consumption order (model ctor, then CP init)."""
    torch.manual_seed(global_seed)
    vit = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=drop_path_rate, depth=depth,
                       num_classes=num_classes, img_size=img_size)
    seeded_backbone_into(vit, backbone_seed)
    cp = O.init_cp_params(rank, l_mu, l_std)
    randomise_cp(cp, cp_seed)
    return O.vit_weights(vit), cp


def oracle_case_cp_length(cp_length):
    """Case 7 of make_golden.py (image_classification/dim_experiment.py's set_CP with cp_length 3 / 5): depth 2, 197
    tokens, rank 16 -- the same global-RNG order (model ctor under seed 14, CP init under seed 15), the same
    non-zero fill of the zero-initialised factors.  Returns (weights, cp, images)."""
    torch.manual_seed(14)
    vit = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=2, num_classes=100)
    seeded_backbone_into(vit, 401)
    torch.manual_seed(15)
    cp = O.init_cp_params(16, 1.5, 0.1, cp_length=cp_length)
    g = torch.Generator(device="cpu").manual_seed(402)
    with torch.no_grad():
        cp["CP_A3" if cp_length == 5 else "CP_A2"].copy_(0.05 * torch.randn(768, 16, generator=g))
        cp["CP_P2"].copy_(0.05 * torch.randn(768, 16, generator=g))
        for k in ("CP_bias1", "CP_bias2", "CP_bias3"):
            cp[k].copy_(0.02 * torch.randn(cp[k].shape, generator=g))
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator(device="cpu").manual_seed(403))
    return O.vit_weights(vit), cp, img


def oracle_case_cp_length2():
    """Case 8 of make_golden.py: order 2 (CP_A2 [dim * dim, rank]), depth 2, 197 tokens, RANK 4 -- same RNG order and fills as
    the script was run with.  Returns (weights, cp, images)."""
    torch.manual_seed(14)
    vit = O.create_vit("vit_base_patch16_224_in21k", drop_path_rate=0.1, depth=2, num_classes=100)
    seeded_backbone_into(vit, 401)
    torch.manual_seed(15)
    cp = O.init_cp_params(4, 1.5, 0.1, cp_length=2)
    g = torch.Generator(device="cpu").manual_seed(402)
    with torch.no_grad():
        cp["CP_A2"].copy_(0.02 * torch.randn(768 * 768, 4, generator=g))
        cp["CP_P2"].copy_(0.05 * torch.randn(768, 4, generator=g))
        for k in ("CP_bias1", "CP_bias2", "CP_bias3"):
            cp[k].copy_(0.02 * torch.randn(cp[k].shape, generator=g))
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator(device="cpu").manual_seed(403))
    return O.vit_weights(vit), cp, img
