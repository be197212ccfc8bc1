"""width/8 CIFAR-shaped architecture dict used by the full-step golden fixture (test infrastructure)."""
import numpy as np


def tiny_architecture(loss_base=4):
    ak = float(np.power(64.0, 0.125))
    s = 's'
    return {'input': [(3, 32, 32)], 'code': [(32, 'linear')],
            'generator': [{'name': 'l1', 'out': 64 * 4 * 4, 'op': 'd', 'act': 'linear', 'act_nm': None,
                           'out_reshape': [64, 4, 4]},
                          {'name': 'l2_up', 'out': 32, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4, 'strides': 2},
                          {'name': 'l3_up', 'out': 16, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4, 'strides': 2},
                          {'name': 'l4_up', 'out': 8, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4, 'strides': 2},
                          {'name': 'l5_t32', 'out': 3, 'act': 'tanh'}],
            'discriminator': [{'name': 'l1_f32', 'out': 8, 'act': 'lrelu', 'act_k': ak, 'w_nm': s},
                              {'name': 'l2_ds', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': s, 'kernel': 4, 'strides': 2},
                              {'name': 'l3', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': s},
                              {'name': 'l4_ds', 'out': 32, 'act': 'lrelu', 'act_k': ak, 'w_nm': s, 'kernel': 4, 'strides': 2},
                              {'name': 'l5', 'out': 32, 'act': 'lrelu', 'act_k': ak, 'w_nm': s},
                              {'name': 'l6_ds', 'out': 64, 'act': 'lrelu', 'act_k': ak, 'w_nm': s, 'kernel': 4, 'strides': 2},
                              {'name': 'l7', 'out': 64, 'op': 'c', 'act': 'lrelu', 'act_k': ak, 'w_nm': s,
                               'out_reshape': [4 * 4 * 64]},
                              {'name': 'l8_s', 'out': 16, 'op': 'd', 'act_k': ak, 'bias': 'b', 'w_nm': s}]}


def tiny_gsn_architecture():
    """the width/8 pair again with spectral norm in the GENERATOR as well: on its dense layer, on transposed-conv kernels
    (SpectralNorm's 'tc' branch, math_func.py:512-528: the power iteration runs on the conv the layer is the transpose of)
    with and without batch norm behind them, and on its last conv - every kernel kind a DCGAN generator has.  D is a
    shortened tiny_architecture()."""
    ak = float(np.power(64.0, 0.125))
    s = 's'
    return {'input': [(3, 16, 16)], 'code': [(24, 'linear')],
            'generator': [{'name': 'l1', 'out': 32 * 4 * 4, 'op': 'd', 'act': 'linear', 'act_k': 1.0, 'w_nm': s,
                           'out_reshape': [32, 4, 4]},
                          {'name': 'l2_up', 'out': 16, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'act_k': ak, 'w_nm': s,
                           'kernel': 4, 'strides': 2},
                          {'name': 'l3_up', 'out': 8, 'op': 'tc', 'act': 'lrelu', 'act_k': ak, 'w_nm': s, 'kernel': 4, 'strides': 2},
                          {'name': 'l4_t16', 'out': 3, 'act': 'tanh', 'act_k': 1.0, 'w_nm': s}],
            'discriminator': [{'name': 'l1_f16', 'out': 8, 'act': 'lrelu', 'act_k': ak, 'w_nm': s},
                              {'name': 'l2_ds', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': s, 'kernel': 4, 'strides': 2},
                              {'name': 'l3', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': s},
                              {'name': 'l4_ds', 'out': 32, 'act': 'lrelu', 'act_k': ak, 'w_nm': s, 'kernel': 4, 'strides': 2,
                               'out_reshape': [4 * 4 * 32]},
                              {'name': 'l5_s', 'out': 16, 'op': 'd', 'act_k': ak, 'bias': 'b', 'w_nm': s}]}


def tiny_res_architecture():
    """width/16 ResNet-SN shaped pair: G = dense -> two up-sampling residual blocks with BN ->
    BN/relu -> conv/tanh at 16x16; D = the 'optimised' first block (res_v1), a down-sampling block, an identity-
    shortcut block, dense - every kind of block the reference defines (layer_func.py:1687-1842)."""
    ak = float(np.power(64.0, 0.125))
    k = [3, 3, 1]
    return {'input': [(3, 16, 16)], 'code': [(24, 'linear')],
            'generator': [{'name': 'l1', 'out': 32 * 4 * 4, 'op': 'd', 'out_reshape': [32, 4, 4]},
                          {'name': 'l2_res', 'type': 'res', 'out': 16, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                           'scale': ['unpool', 2]},
                          {'name': 'l3_res', 'type': 'res', 'out': 16, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                           'scale': ['unpool', 2]},
                          {'name': 'l4_bn', 'op': 'i', 'act': 'relu', 'act_nm': 'bn'},
                          {'name': 'l5_t16', 'out': 3, 'act': 'tanh'}],
            'discriminator': [{'name': 'l1_res', 'type': 'res_v1', 'out': 16, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                               'kernel': k, 'scale': ['avg', -2]},
                              {'name': 'l2_res', 'type': 'res', 'out': 32, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                               'kernel': k, 'scale': ['avg', -2]},
                              {'name': 'l3_res', 'type': 'res_i', 'out': 32, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                               'out_reshape': [4 * 4 * 32]},
                              {'name': 'l4_s', 'out': 16, 'op': 'd', 'act_k': ak, 'w_nm': 's'}]}


def tiny_res_ps_architecture():
    """the residual pair again with periodic shuffling ('ps', layer_func.py:197-244) as the scaling method: up-sampling
    moves 4 channels into a 2x2 block (32 -> 8 channels in front of kernel_0 and kernel_sc), down-sampling the reverse"""
    ak = float(np.power(64.0, 0.125))
    k = [3, 3, 1]
    return {'input': [(3, 16, 16)], 'code': [(24, 'linear')],
            'generator': [{'name': 'l1', 'out': 32 * 4 * 4, 'op': 'd', 'out_reshape': [32, 4, 4]},
                          {'name': 'l2_res', 'type': 'res', 'out': 32, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                           'scale': ['ps', 2]},
                          {'name': 'l3_res', 'type': 'res', 'out': 16, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                           'scale': ['ps', 2]},
                          {'name': 'l4_t16', 'out': 3, 'act': 'tanh'}],
            # (a res_v1 block cannot shuffle down: its shortcut shuffles BEFORE the 1x1 kernel, the branch after - the
            # reference's own shape check rejects it, layer_func.py:1767)
            'discriminator': [{'name': 'l1_res', 'type': 'res', 'out': 8, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's',
                               'kernel': k, 'scale': ['ps', -2]},
                              {'name': 'l2_res', 'type': 'res', 'out': 8, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's',
                               'kernel': k, 'scale': ['ps', -2], 'out_reshape': [4 * 4 * 32]},
                              {'name': 'l3_s', 'out': 16, 'op': 'd', 'act_k': ak, 'w_nm': 's'}]}


def tiny_res_bil_architecture():
    """the residual pair with bilinear resizing ('bil', tf.image.resize_bilinear align_corners=True, layer_func.py:
    1128-1137) as the scaling method, both directions; 12x12 images so that the interpolation weights are not all
    dyadic (4 -> 6 -> 12 up with factors 1.5 is not expressible: factors are integers, so 3 -> 6 -> 12)"""
    ak = float(np.power(64.0, 0.125))
    k = [3, 3, 1]
    return {'input': [(3, 12, 12)], 'code': [(24, 'linear')],
            'generator': [{'name': 'l1', 'out': 16 * 3 * 3, 'op': 'd', 'out_reshape': [16, 3, 3]},
                          {'name': 'l2_res', 'type': 'res', 'out': 16, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                           'scale': ['bil', 2]},
                          {'name': 'l3_res', 'type': 'res', 'out': 8, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                           'scale': ['bil', 2]},
                          {'name': 'l4_t12', 'out': 3, 'act': 'tanh'}],
            'discriminator': [{'name': 'l1_res', 'type': 'res_v1', 'out': 8, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                               'kernel': k, 'scale': ['bil', -2]},
                              {'name': 'l2_res', 'type': 'res', 'out': 16, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                               'kernel': k, 'scale': ['bil', -3], 'out_reshape': [2 * 2 * 16]},
                              {'name': 'l3_s', 'out': 16, 'op': 'd', 'act_k': ak, 'w_nm': 's'}]}


def tiny_res_bic_architecture():
    """the same pair with bicubic resizing ('bic', tf.image.resize_bicubic align_corners=True, layer_func.py:1138-1147):
    x2 twice in G, /2 and /3 in D - taps that run off the image (clamped) on every border"""
    arch = tiny_res_bil_architecture()
    for net in ('generator', 'discriminator'):
        for layer in arch[net]:
            if 'scale' in layer:
                layer['scale'] = ['bic', layer['scale'][1]]
    return arch


def tiny_res_max_architecture():
    """max pooling ('max', layer_func.py:1149-1153) as the down-sampling method, and scaling on PLAIN layers as well
    (layer_func.py:1627-1642: up-sampling in front of the kernel, down-sampling behind the activation): G's second
    stage is a plain conv layer with 'unpool', D's first a plain SN conv layer with 'max'"""
    ak = float(np.power(64.0, 0.125))
    k = [3, 3, 1]
    return {'input': [(3, 16, 16)], 'code': [(24, 'linear')],
            'generator': [{'name': 'l1', 'out': 32 * 4 * 4, 'op': 'd', 'out_reshape': [32, 4, 4]},
                          {'name': 'l2_res', 'type': 'res', 'out': 16, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                           'scale': ['unpool', 2]},
                          {'name': 'l3_up', 'out': 16, 'act': 'relu', 'act_nm': 'bn', 'scale': ['unpool', 2]},
                          {'name': 'l4_t16', 'out': 3, 'act': 'tanh'}],
            'discriminator': [{'name': 'l1_ds', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's', 'scale': ['max', -2]},
                              {'name': 'l2_res', 'type': 'res', 'out': 32, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                               'kernel': k, 'scale': ['max', -2]},
                              {'name': 'l3_res', 'type': 'res_i', 'out': 32, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                               'out_reshape': [4 * 4 * 32]},
                              {'name': 'l4_s', 'out': 16, 'op': 'd', 'act_k': ak, 'w_nm': 's'}]}


def tiny_res_tc_architecture():
    """residual blocks built on TRANSPOSED convolutions (op 'tc' inside a block, layer_func.py:1725-1727: kernel_0 and the
    shortcut's kernel_sc are transposed convs - they do the up-sampling, 'scale' is dropped for 'tc', :1245-1247 - and
    kernel_1 is a conv): G = dense -> a 4x4/2 block with BN (4x4/2 shortcut) -> a spectrally normalised block whose
    shortcut is a 1x1/2 transposed conv (a pixel every other position) -> an identity-shortcut block on 3x3/1 transposed
    convs -> conv/tanh at 16x16.  'kernel' / 'strides' as per-kernel lists (Layer._update_design_, :1380-1395).
    D is tiny_res_architecture()'s."""
    ak = float(np.power(64.0, 0.125))
    arch = tiny_res_architecture()
    arch['generator'] = [{'name': 'l1', 'out': 32 * 4 * 4, 'op': 'd', 'out_reshape': [32, 4, 4]},
                         {'name': 'l2_res', 'type': 'res', 'op': 'tc', 'out': 16, 'act': 'relu', 'act_nm': 'bn',
                          'kernel': [4, 3, 4], 'strides': [2, 1, 2]},
                         {'name': 'l3_res', 'type': 'res', 'op': 'tc', 'out': 8, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's',
                          'kernel': [4, 3, 1], 'strides': [2, 1, 2]},
                         {'name': 'l4_res', 'type': 'res_i', 'op': 'tc', 'out': 8, 'act': 'relu', 'act_nm': 'bn',
                          'kernel': 3, 'strides': 1},
                         {'name': 'l5_t16', 'out': 3, 'act': 'tanh'}]
    return arch
