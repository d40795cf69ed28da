"""CPU: every kernel variant that BASELINE configs 2, 3 and 4 run has a parity case on the GPU (tests/test_gpu_layers.py).

The kernels pick a template instantiation per launch (channel panel, voxel tile, weights in LDS or L2, fused / class-parallel
output-parity classes, single-channel staging, the 32x32x16 flavour, weight-gradient slab shapes ...), and the choice depends
on the problem size -- so a parity suite made of small hand-picked shapes can be green while the kernels the benchmark
times are never compared with the oracle.  Here the real schedules are walked in dry-run mode (tests/layer_recipes.py) and
the variant of every convolution launch is collected through vg_conv3d_variant / vg_conv3d_wgrad_variant."""
import pytest

import layer_recipes as LR


def test_every_variant_of_the_baseline_configs_has_a_gpu_parity_case():
    reps = LR.representatives()
    needed = LR.needed_variants()
    assert len(needed) >= 60                       # sanity: the walk saw the whole step
    missing = [kv for kv in needed if kv not in reps]
    assert not missing, missing
    # the GPU test is parametrised over exactly these representatives; each is a call of a real layer at its true shape
    for kv in needed:
        r = reps[kv]
        assert r['config'] in LR.CONFIGS and r['recipe']['kind'] == kv[0]
    # kinds of kernels that must be present: both MFMA flavours, fused and class-parallel strided data gradients, the
    # single-channel staging variant, the pointwise kernels, the three weight-gradient slab families (the DMA one with the materialised operand, d0, and copying the stored tensors, d1), the K-split deep-level launches (ks), the thin-channel specialist
    # (both staging modes) and the shortcut-convolution kernels (forward with geometry, dense and strided accumulating data gradient, the DMA weight gradient
    # of the 1x1x1 layers in every instantiation, with the virtual upsample + concat and with stride 2)
    names = ' '.join(v for _, v in needed)
    for frag in ('conv<bf16,16,8', 'conv<bf16,32,4', 'conv32<128', 'conv32<64', 'mc1', 'mc2', 'cp1', 'c11', 'pw_cto1', 'pw_1toc',
                 'c1m_fwd<f32,3,1,1,n0>', 'c1m_wgrad<f32,3,1,1,n0>', 'c1m_fwd<f32,4,2,4,n1>', 'c1m_wgrad<f32,4,2,4,n1>', 'wgrad<bf16,24,1', 'wgrad<bf16,7,4', 'walk1', 'ch1', 'part1', 'n1',
                 'wgrad_dma<4,1,d0>', 'wgrad_thin<m1>|ch0', 'wgrad_thin<m1>|ch1', 'wgrad_dma<4,2,d0>', 'wgrad_dma<4,2,d1>', 'wgrad_dma<8,2,d0>', 'wgrad_dma<8,4,d0>', '|s2|', '|nb3|', '|ks4', '|ks2',
                 'conv_thin<m0', 'conv_thin<m1', 'pw_gemm<2,1,g1,a0>', 'pw_gemm<1,3,g0,a1>', 'pw_gemm<1,1,g1,a1>',
                 'wgrad_pw_dma<3,1>', 'wgrad_pw_dma<6,2>', 'wgrad_pw_dma<4,4>', 'wgrad_pw_dma<6,4>', '|s1|u1|', '|s2|u0|',
                 # the LDS-DMA forward / data-gradient family: both panel widths, 4^3 and 3^3 stage bodies, K split, class-parallel strided
                 # data gradients, launches with more units than CUs
                 # the two-panel instance of the thin-channel specialist (32-channel layers at 64^3), forward and data gradient
                 'conv_thin2<m1', 'conv_thin2<m0',
                 # ... and its panel loop (dec0.cb1's 16 -> 48 data gradient with the IN-backward statistics: three panels over one staged halo)
                 'bs2,pl>',
                 'conv_dma<128,256>', 'conv_dma<64,256>', '|gt8|', '|gt9|', 'ks1|cls0', 'ks0|cls1', 'cls1|walk1'):
        assert frag in names, frag


def test_resnet_generator_walk():
    """the kernel variants the ResNet generator launches at the 128^3 patch are all exercised by its teacher-forced GPU
    tests (tests/test_gpu_resnet.py runs the whole network, forward and backward, at 32^3 batch 2, 64^3 and 128^3 -- the walk below
    is what those runs launch), and the families one expects are there: the W-packed 7^3 stem, the seven 49-tap chunks of the head,
    class-parallel strided data gradients on odd grids, the LDS-DMA data gradients and weight gradients of the 256-channel trunk."""
    sizes = {'32^3 B2': ((32, 32, 32), 2), '64^3 B1': ((64, 64, 64), 1), '128^3 B1': ((128, 128, 128), 1)}
    recs = {k: LR.enumerate_resnet(*v) for k, v in sizes.items()}
    var = {k: {(kind, v) for kind, _, v in r} for k, r in recs.items()}
    assert len(recs['128^3 B1']) == 77                      # 26 convolutions: 32 forward launches (head: 7), 32 + 13 backward
    only_128 = var['128^3 B1'] - var['64^3 B1'] - var['32^3 B2']
    assert len(only_128) >= 4, sorted(only_128)             # why the 128^3 GPU test exists (13 variants no smaller grid selects)
    names = ' '.join(v for k in var for _, v in var[k])
    for frag in ('c11', 'conv32<128', 'conv32<64', 'cp1', 'mc1', 'conv_dma<128,256>', 'wgrad_dma<8,2,d0>', 'wgrad_dma<8,4,d0>', 'wgrad<bf16,7,4'):
        assert frag in names, frag
    heads = [n for kind, n, _ in recs['128^3 B1'] if kind == 'fwd' and n == 'out']
    assert len(heads) == 7


def test_walk_is_complete():
    """One generator application is 29 forward + 29 weight-gradient + 28 data-gradient calls (the stem's shortcut is no convolution
    launch in either direction: forward an affine function of the volume added in stem.cb's epilogue, vg_stem_short_fwd / res_c1,
    backward a closed form in two moments, vg_stem_short_bwd; stem.conv1 has no data gradient), one discriminator 5 + 5 + 2 x 5 (the
    second sweep stops at the input volume)."""
    recs = LR.all_records()['32^3 B1']
    kinds = [k for k, _, _, _ in recs]
    names = {n for _, n, _, _ in recs}
    assert kinds.count('fwd') == 34 and kinds.count('wgrad') == 34
    assert {'stem.conv1', 'dec0.cb1.conv', 'bridge.cb2.conv', 'out', 'conv0', 'down2'} <= names


@pytest.mark.parametrize('cfg', LR.NEEDED)
def test_variant_names_are_stable_within_a_config(cfg):
    """The same call selects the same variant when asked twice (the dispatch has no hidden state)."""
    dims, B = LR.CONFIGS[cfg]
    a = LR.enumerate_config(dims, B)
    assert [(k, n, v) for k, n, v, _ in a] == [(k, n, v) for k, n, v, _ in LR.all_records()[cfg]]
