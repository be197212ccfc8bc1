"""Architecture dicts and hyper-parameters of the reference's four driver scripts, restated
(my_test_cifar.py:9-56, my_test_stl.py:7-50, my_test_celebA.py:8-56, my_test_lsun.py:8-56)."""
import numpy as np


def _dcgan(base, g_first_bn, n_stage, image, act_k, width=64):
    """DCGAN-SN generator/discriminator pair: `n_stage` up/down-sampling stages."""
    w_nm = 's'
    top = width * 2 ** n_stage                      # 512 (3 stages) / 1024 (4 stages)
    gen = [{'name': 'l1', 'out': top * base * base, 'op': 'd', 'out_reshape': [top, base, base],
            **({'act': 'relu', 'act_nm': 'bn'} if g_first_bn else {'act': 'linear', 'act_nm': None})}]
    ch = top
    for i in range(n_stage):
        ch //= 2
        gen.append({'name': 'l{}_up'.format(i + 2), 'out': ch, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4,
                    'strides': 2})
    gen.append({'name': 'l{}_t32'.format(n_stage + 2), 'out': 3, 'act': 'tanh'})     # the scripts keep 't32'/'f32' at every size
    dis = [{'name': 'l1_f32', 'out': width, 'act': 'lrelu', 'act_k': act_k, 'w_nm': w_nm}]
    ch, idx = width, 2
    for i in range(n_stage):
        dis.append({'name': 'l{}_ds'.format(idx), 'out': ch * 2, 'act': 'lrelu', 'act_k': act_k, 'w_nm': w_nm, 'kernel': 4,
                    'strides': 2})
        ch *= 2
        idx += 1
        last = i == n_stage - 1
        layer = {'name': 'l{}'.format(idx), 'out': ch, 'act': 'lrelu', 'act_k': act_k, 'w_nm': w_nm}
        if last:
            layer.update({'op': 'c', 'out_reshape': [base * base * ch]})
        dis.append(layer)
        idx += 1
    dis.append({'name': 'l{}_s'.format(idx), 'out': 16, 'op': 'd', 'act_k': act_k, 'bias': 'b', 'w_nm': w_nm})
    return {'input': [(3, image, image)], 'code': [(128, 'linear')], 'generator': gen, 'discriminator': dis}


def cifar():
    """my_test_cifar.py:9-38: 32x32, 8 SN layers in D, act_k = 64^(1/8), lr (D,G) = 5e-4 / 2e-4"""
    return _dcgan(4, False, 3, 32, float(np.power(64.0, 0.125))), [5e-4, 2e-4]


def stl():
    """my_test_stl.py:7-32: 48x48 (base 6), G l1 = dense + BN + relu, lr 2e-4 / 2e-4"""
    return _dcgan(6, True, 3, 48, float(np.power(64.0, 0.125))), [2e-4, 2e-4]


def celeba():
    """my_test_celebA.py:8-38: 64x64, 10 SN layers in D, act_k = 64^(1/10), lr 1e-4 / 2e-4"""
    return _dcgan(4, False, 4, 64, float(np.power(64.0, 0.1))), [1e-4, 2e-4]


def lsun():
    """my_test_lsun.py:8-38: same nets as CelebA, lr 2e-4 / 1e-4"""
    return _dcgan(4, False, 4, 64, float(np.power(64.0, 0.1))), [2e-4, 1e-4]


def lsun_resnet():
    """BASELINE.json config 5 (LSUN-bedroom 64x64 ResNet-SN).  The reference ships the residual-block layer type
    (layer_func.py:1687-1842) but no driver script with a ResNet dict; this one is authored from its block definitions
    in the usual SN-GAN 64x64 layout: G = dense -> [1024,4,4] -> four up-sampling blocks (BN, relu, 'unpool' x2,
    3x3 / 3x3 / 1x1 shortcut) -> BN-relu -> conv-tanh; D = the 'optimised' first block (res_v1), three down-sampling
    blocks ('avg' /2), one identity-shortcut block, dense to the 16-d score; every D kernel spectrally normalised with
    act_k = 64^(1/10) as in my_test_lsun.py:9.  lr as my_test_lsun.py."""
    ak = float(np.power(64.0, 0.1))
    k = [3, 3, 1]
    gen = [{'name': 'l1', 'out': 1024 * 4 * 4, 'op': 'd', 'out_reshape': [1024, 4, 4]}]
    for i, ch in enumerate((512, 256, 128, 64)):
        gen.append({'name': 'l{}_res'.format(i + 2), 'type': 'res', 'out': ch, 'act': 'relu', 'act_nm': 'bn', 'kernel': k,
                    'scale': ['unpool', 2]})
    gen += [{'name': 'l6_bn', 'op': 'i', 'act': 'relu', 'act_nm': 'bn'}, {'name': 'l7_t64', 'out': 3, 'act': 'tanh'}]
    dis = [{'name': 'l1_res', 'type': 'res_v1', 'out': 64, 'act': 'relu', 'act_k': ak, 'w_nm': 's', 'kernel': k,
            'scale': ['avg', -2]}]
    for i, ch in enumerate((128, 256, 512)):
        dis.append({'name': 'l{}_res'.format(i + 2), 'type': 'res', 'out': ch, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
                    'kernel': k, 'scale': ['avg', -2]})
    dis += [{'name': 'l5_res', 'type': 'res_i', 'out': 512, 'act': 'relu', 'act_k': ak, 'w_nm': 's',
             'out_reshape': [4 * 4 * 512]},
            {'name': 'l6_s', 'out': 16, 'op': 'd', 'act_k': ak, 'bias': 'b', 'w_nm': 's'}]
    return {'input': [(3, 64, 64)], 'code': [(128, 'linear')], 'generator': gen, 'discriminator': dis}, [2e-4, 1e-4]


CONFIGS = {'cifar': cifar, 'stl': stl, 'celeba': celeba, 'lsun': lsun, 'lsun_resnet': lsun_resnet}


def flops_per_image(architecture):
    """forward 2*MAC per image of G and D; step FLOPs = B*(3 F_G + 7 F_D)."""
    def net(layers, shape):
        total = 0
        for d in layers:
            op = d.get('op', 'c')
            k, s = d.get('kernel', 3), d.get('strides', 1)
            if d.get('type', 'default') in ('res', 'res_i', 'res_v1'):         # layer_func.py:1687-1842
                c, h, w = shape
                ks = k if isinstance(k, (list, tuple)) else [k, k, k]
                f = d['scale'][1] if d.get('scale') else 1
                hu, wu = (h * f, w * f) if f > 0 else (h, w)                     # the 3x3 convs run at the larger size
                total += 2 * ks[0] ** 2 * c * d['out'] * hu * wu + 2 * ks[1] ** 2 * d['out'] * d['out'] * hu * wu
                ho, wo = (hu, wu) if f > 0 else (h // -f, w // -f)
                if d['type'] == 'res':
                    total += 2 * ks[2] ** 2 * c * d['out'] * hu * wu
                elif d['type'] == 'res_v1':
                    total += 2 * ks[2] ** 2 * c * d['out'] * ho * wo             # the shortcut convolves the pooled input
                shape = d.get('out_reshape', [d['out'], ho, wo])
            elif op == 'i':
                shape = d.get('out_reshape', shape)
            elif op == 'd':
                total += 2 * int(np.prod(shape)) * d['out']
                shape = d.get('out_reshape', [d['out']])
            elif op == 'c':
                c, h, w = shape
                ho, wo = -(-h // s), -(-w // s)
                total += 2 * k * k * c * d['out'] * ho * wo
                shape = d.get('out_reshape', [d['out'], ho, wo])
            else:
                c, h, w = shape
                total += 2 * k * k * c * d['out'] * h * w        # every input pixel meets every tap once
                shape = [d['out'], h * s, w * s]
        return total
    fg = net(architecture['generator'], [architecture['code'][0][0]])
    fd = net(architecture['discriminator'], list(architecture['input'][0]))
    return fg, fd
