"""CPU: the oracle's fp16 variant (oracle/gmfnet_ref.py, cfg['gmf']['half']) against its own fp32 form on pre-rounded operands."""
def test_half_variant_of_the_oracle_is_fp32_arithmetic_on_rounded_operands():
    """cfg['gmf']['half'] = 1 (the statement the fp16 HIP kernels are held to, tests/test_gpu_half.py): spec_a sees its input
    and weight rounded to fp16 (nearest even), everything else is the fp32 network; the roundings pass gradients straight
    through, so the gradient of the fp32 master weight is the gradient w.r.t. its rounded copy."""
    import torch
    from oracle.gmfnet_ref import Net
    cfg = {'patch_size': 5, 'Categories_Number': 5, 'data_city': 's', 'DATA_DICT': {'s': {'size': [16, 16, 8]}}, 'scale': 1,
           'aux_bands': 1, 'gmf': {'width': 40, 'half': 1}}
    torch.manual_seed(0)
    net = Net(cfg)
    cfg32 = dict(cfg, gmf={'width': 40, 'half': 0})
    ref = Net(cfg32)
    ref.load_state_dict(net.state_dict())
    with torch.no_grad():
        ref.spec_a.weight.copy_(ref.spec_a.weight.to(torch.float16).float())
    a = torch.randn(7, 8, 5, 5) * 3
    a[0, 0, 0, 0] = 3e-5                      # fp16-subnormal magnitude: kept, not flushed
    b = torch.rand(7, 1, 5, 5)
    t = torch.randint(0, 5, (7,))
    out = net(a, b)
    want = ref(a.to(torch.float16).float(), b)
    assert torch.equal(out, want)
    torch.nn.functional.cross_entropy(out, t).backward()
    torch.nn.functional.cross_entropy(want, t).backward()
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p.grad, q.grad), k
