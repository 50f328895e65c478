"""ctypes binding of libdmf_hip.so (include/dmf.h).  PyTorch is used only for device memory and streams.

The library is mandatory: importing this module without the built .so raises (there is NO CPU fallback —
the CPU statement of the arithmetic lives under oracle/ and is test infrastructure only).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DMF_LIB', os.path.join(_HERE, 'libdmf_hip.so'))   # DMF_LIB: diagnostic builds only
KMAX = 64


class DmfError(RuntimeError):
    pass


class Shape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('C', 'C2', 'P', 'S', 'F', 'G', 'H', 'K', 'attention', 'heads', 'E', 'reserved')]


class XgmiComm(C.Structure):
    _fields_ = [('world', C.c_int32), ('rank', C.c_int32), ('capacity', C.c_int64), ('timeout_ms', C.c_int32),
                ('seq_bias', C.c_int32), ('data', C.c_void_p * 16), ('flags', C.c_void_p * 16)]


class QuaParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('alpha', 'beta', 'gamma', 'epsilon', 'tao')]


class Input(C.Structure):
    _fields_ = [('mode', C.c_int32), ('B', C.c_int32), ('a', C.c_void_p), ('b', C.c_void_p), ('sceneA', C.c_void_p),
                ('sceneB', C.c_void_p), ('xy', C.c_void_p), ('Wp', C.c_int32), ('WpB', C.c_int32), ('cursor', C.c_void_p),
                ('half', C.c_int32), ('reserved', C.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise DmfError('libdmf_hip.so is not built: run `python dual-modal-fusion_amd/build.py` '
                       '(hipcc --offload-arch=gfx950); there is no CPU fallback for the product path')
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    SP, IP = C.POINTER(Shape), C.POINTER(Input)
    protos = {
        'dmf_version': (i32, []),
        'dmf_last_error': (C.c_char_p, []),
        'dmf_shape_supported': (i32, [SP]),
        'dmf_patch_variant': (i32, [SP, i32]),
        'dmf_param_layout': (i32, [SP, C.POINTER(i64)]),
        'dmf_workspace_bytes': (i64, [SP, i32]),
        'dmf_forward': (i32, [SP, IP, vp, vp, vp, vp, vp]),
        'dmf_attn_workspace_bytes': (i64, [SP, i32]),
        'dmf_forward_attn': (i32, [SP, IP, vp, vp, vp, vp, vp, vp]),
        'dmf_train_fwd_bwd': (i32, [SP, IP, vp, vp, vp, f32, vp, vp, vp, vp, vp]),
        'dmf_train_fwd_bwd_scaled': (i32, [SP, IP, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp]),
        'dmf_half_supported': (i32, [SP]),
        'dmf_scaler_init': (i32, [vp, f32, vp]),
        'dmf_unscale_adam': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, f32, f32, i32, i32, vp, vp, vp]),
        'dmf_grad_reduce_scaled': (i32, [SP, i32, vp, vp, vp, vp, vp, vp, vp]),
        'dmf_qua_loss_scaled': (i32, [vp, i32, i32, vp, vp, C.POINTER(QuaParams), f32, vp, vp, vp, vp, vp]),
        'dmf_attn_train_workspace_bytes': (i64, [SP, i32]),
        'dmf_train_attn_fwd_bwd': (i32, [SP, IP, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp]),
        'dmf_backward_dlogits': (i32, [SP, IP, vp, vp, vp, vp, vp]),
        'dmf_unit_supported': (i32, [SP]),
        'dmf_forward_unit': (i32, [SP, IP, vp, vp, vp, vp, vp, vp]),
        'dmf_backward_unit': (i32, [SP, i32, vp, vp, vp, vp]),
        'dmf_grad_reduce': (i32, [SP, i32, vp, vp, vp]),
        'dmf_adam_step': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, vp, vp, vp]),
        'dmf_sgd_step': (i32, [vp, vp, vp, i64, f32, f32, i32, f32, vp, vp, vp]),
        'dmf_rmsprop_step': (i32, [vp, vp, vp, i64, f32, f32, f32, f32, vp, vp]),
        'dmf_grad_reduce_adam': (i32, [SP, i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, i32, vp, vp, vp, vp, vp]),
        'dmf_forward_ce': (i32, [SP, IP, vp, vp, vp, vp, vp, vp, vp]),
        'dmf_train_plan_steps': (i32, [SP, IP, vp, vp, vp, f32, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp, vp, vp, i32, vp]),
        'dmf_xgmi_sizes': (i32, [i64, i32, C.POINTER(i64), C.POINTER(i64)]),
        'dmf_xgmi_alloc': (i32, [i64, C.POINTER(vp)]),
        'dmf_xgmi_free': (i32, [vp]),
        'dmf_xgmi_export': (i32, [vp, C.c_char_p]),
        'dmf_xgmi_open': (i32, [C.c_char_p, C.POINTER(vp)]),
        'dmf_xgmi_close': (i32, [vp]),
        'dmf_xgmi_status': (i32, [C.POINTER(XgmiComm), C.POINTER(i32)]),
        'dmf_xgmi_allreduce': (i32, [C.POINTER(XgmiComm), vp, i64, i32, vp]),
        'dmf_grad_reduce_xgmi_adam': (i32, [SP, i32, vp, vp, vp, vp, C.POINTER(XgmiComm), f32, f32, f32, f32, f32, vp, vp,
                                            vp, vp, vp]),
        'dmf_qua_loss': (i32, [vp, i32, i32, vp, vp, C.POINTER(QuaParams), f32, vp, vp, vp, vp]),
        'dmf_pair_argmax': (i32, [vp, i32, i32, vp, vp]),
        'dmf_band_mean': (i32, [vp, i32, i64, i64, i32, vp, vp]),
        'dmf_confusion_accum': (i32, [vp, vp, i32, i32, vp, vp]),
        'dmf_labelmap_write': (i32, [vp, vp, i32, i32, vp, vp]),
        'dmf_pan2ms': (i32, [vp, i32, i32, i32, vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)       # AttributeError here = header and library disagree
        fn.restype, fn.argtypes = res, args
    return lib, tuple(protos)


_lib, EXPORTS = _load()


def check(rc):
    if rc != 0:
        raise DmfError(_lib.dmf_last_error().decode() or 'libdmf_hip error %d' % rc)


def version():
    return _lib.dmf_version()


def make_shape(arch):
    return Shape(C=arch['C'], C2=arch['C2'], P=arch['P'], S=arch['S'], F=arch['F'], G=arch['G'], H=arch['H'],
                 K=arch['K'], attention=arch.get('attention', 0), heads=arch.get('heads', 0), E=arch.get('E', 0), reserved=0)


def shape_supported(shape):
    check(_lib.dmf_shape_supported(C.byref(shape)))


def patch_v2_used(shape, mode=1):
    """True if the train step (mode 1; 0 = forward, 2 = backward from dlogits) of this shape runs the v2 patch kernel."""
    return _lib.dmf_patch_variant(C.byref(shape), mode) == 2


def unit_supported(shape):
    """True if the two-launch unit-gradient step (forward_unit / backward_unit) exists for this shape."""
    return _lib.dmf_unit_supported(C.byref(shape)) == 0


def half_supported(shape):
    """True if the fp16-scene kernels (Input.half) exist for this shape."""
    return _lib.dmf_half_supported(C.byref(shape)) == 0


def require_half(shape):
    check(_lib.dmf_half_supported(C.byref(shape)))


SCALER_FLOATS = 8


def scaler_init(state, init_scale):
    _dev(state, torch.float32, 'scaler state')
    if state.numel() < SCALER_FLOATS:
        raise DmfError('scaler state needs %d floats' % SCALER_FLOATS)
    check(_lib.dmf_scaler_init(_ptr(state), init_scale, _stream()))


def unscale_adam(theta, grad, m, v, lr, b1, b2, eps, scaler_state, growth_factor, backoff_factor, growth_interval,
                 adam_step_dev, grad_scale=1.0, cursor_dev=None, unscaled=False):
    check(_lib.dmf_unscale_adam(_ptr(theta), _ptr(grad), _ptr(m), _ptr(v), theta.numel(), lr, b1, b2, eps, grad_scale,
                                _ptr(scaler_state), growth_factor, backoff_factor, growth_interval, int(bool(unscaled)),
                                _ptr(adam_step_dev), _ptr(cursor_dev), _stream()))


def grad_reduce_scaled(shape, B, ws, grad, scaler_state, cursor_dev=None, loss=None, loss_hist=None):
    check(_lib.dmf_grad_reduce_scaled(C.byref(shape), B, _ptr(ws), _ptr(grad), _ptr(scaler_state), _ptr(cursor_dev),
                                      _ptr(loss), _ptr(loss_hist), _stream()))


def forward_unit(shape, inp, theta, pool_w, logits, ws, adam_step_dev=None):
    check(_lib.dmf_forward_unit(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(logits), _ptr(ws),
                                _ptr(adam_step_dev), _stream()))


def backward_unit(shape, B, theta, dlogits, ws):
    check(_lib.dmf_backward_unit(C.byref(shape), B, _ptr(theta), _ptr(dlogits), _ptr(ws), _stream()))


def param_layout(shape):
    off = (C.c_int64 * 17)()
    check(_lib.dmf_param_layout(C.byref(shape), off))
    return list(off)


def workspace_bytes(shape, B):
    n = _lib.dmf_workspace_bytes(C.byref(shape), B)
    if n < 0:
        raise DmfError('dmf_workspace_bytes failed')
    return n


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, name):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise DmfError('%s must be a contiguous %s tensor on the GPU' % (name, dtype))
    return t


def input_patches(shape, a, b, half=False):
    """mode 0: the reference dataloader's tensors, a [B,C,P,P], b [B,C2,SP,SP] (dataset.py:168-185).
    half: `a` is rounded to fp16 as the kernel stages it (the arithmetic of an fp16 scene)."""
    _dev(a, torch.float32, 'a'); _dev(b, torch.float32, 'b')
    B = a.shape[0]
    SP = shape.S * shape.P
    if tuple(a.shape) != (B, shape.C, shape.P, shape.P) or tuple(b.shape) != (B, shape.C2, SP, SP):
        raise DmfError('patch tensors %s / %s do not match shape C=%d P=%d C2=%d S=%d' %
                       (tuple(a.shape), tuple(b.shape), shape.C, shape.P, shape.C2, shape.S))
    return Input(mode=0, B=B, a=a.data_ptr(), b=b.data_ptr(), sceneA=None, sceneB=None, xy=None, Wp=0, WpB=0, cursor=None,
                 half=int(bool(half)), reserved=0)


def input_gather(shape, sceneA, sceneB, xy, B=None, cursor=None):
    """mode 1: sceneA [Hp,Wp,C], sceneB [HpB,WpB,C2] resident padded scenes, xy [B,2] int32 top-left pixels.
    With `cursor` (device int32[1]) xy is a whole epoch plan [n_steps*B, 2] and `B` the batch size.
    sceneA may be fp16 (Input.half: shapes of half_supported())."""
    half = sceneA.dtype == torch.float16
    _dev(sceneA, torch.float16 if half else torch.float32, 'sceneA'); _dev(sceneB, torch.float32, 'sceneB'); _dev(xy, torch.int32, 'xy')
    if sceneA.dim() != 3 or sceneA.shape[2] != shape.C or sceneB.dim() != 3 or sceneB.shape[2] != shape.C2:
        raise DmfError('scene tensors must be [Hp,Wp,C] and [HpB,WpB,C2]')
    if xy.dim() != 2 or xy.shape[1] != 2:
        raise DmfError('xy must be [B,2]')
    return Input(mode=1, B=xy.shape[0] if B is None else B, a=None, b=None, sceneA=sceneA.data_ptr(),
                 sceneB=sceneB.data_ptr(), xy=xy.data_ptr(), Wp=sceneA.shape[1], WpB=sceneB.shape[1],
                 cursor=None if cursor is None else cursor.data_ptr(), half=int(half), reserved=0)


def check_xy_bounds(shape, sceneA, sceneB, xy_host):
    """Host-side guard (a faulting gather can reset the GPU): every patch window must lie inside the scenes."""
    if len(xy_host) == 0:
        return
    x_max, y_max = int(xy_host[:, 0].max()), int(xy_host[:, 1].max())
    if int(xy_host.min()) < 0 or x_max + shape.P > sceneA.shape[0] or y_max + shape.P > sceneA.shape[1] or \
            shape.S * (x_max + shape.P) > sceneB.shape[0] or shape.S * (y_max + shape.P) > sceneB.shape[1]:
        raise DmfError('patch window outside the padded scene')


def forward(shape, inp, theta, pool_w, logits, pred=None):
    check(_lib.dmf_forward(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(logits), _ptr(pred), _stream()))


def attn_workspace_bytes(shape, B):
    return _lib.dmf_attn_workspace_bytes(C.byref(shape), B)


def forward_attn(shape, inp, theta, pool_w, ws, logits, pred=None):
    check(_lib.dmf_forward_attn(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(ws), _ptr(logits),
                                _ptr(pred), _stream()))


def train_fwd_bwd(shape, inp, theta, pool_w, labels, loss_scale, logits, loss, ws, adam_step_dev=None, scaler_state=None):
    if scaler_state is not None:
        check(_lib.dmf_train_fwd_bwd_scaled(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(labels),
                                            C.c_float(loss_scale), _ptr(scaler_state), _ptr(logits), _ptr(loss), _ptr(ws),
                                            _ptr(adam_step_dev), _stream()))
        return
    check(_lib.dmf_train_fwd_bwd(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(labels),
                                 C.c_float(loss_scale), _ptr(logits), _ptr(loss), _ptr(ws), _ptr(adam_step_dev), _stream()))


def attn_train_workspace_bytes(shape, B):
    n = _lib.dmf_attn_train_workspace_bytes(C.byref(shape), B)
    if n < 0:
        raise DmfError('dmf_attn_train_workspace_bytes failed')
    return n


def train_attn_fwd_bwd(shape, inp, theta, pool_w, labels, dlogits, loss_scale, logits, loss, ws, attn_ws,
                       adam_step_dev=None):
    check(_lib.dmf_train_attn_fwd_bwd(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(labels), _ptr(dlogits),
                                      loss_scale, _ptr(logits), _ptr(loss), _ptr(ws), _ptr(attn_ws), _ptr(adam_step_dev),
                                      _stream()))


def backward_dlogits(shape, inp, theta, pool_w, dlogits, ws):
    check(_lib.dmf_backward_dlogits(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(dlogits), _ptr(ws), _stream()))


def grad_reduce(shape, B, ws, grad):
    check(_lib.dmf_grad_reduce(C.byref(shape), B, _ptr(ws), _ptr(grad), _stream()))


def adam_step(theta, grad, m, v, lr, b1, b2, eps, step, grad_scale=1.0, adam_step_dev=None, cursor_dev=None):
    check(_lib.dmf_adam_step(_ptr(theta), _ptr(grad), _ptr(m), _ptr(v), theta.numel(), lr, b1, b2, eps, step,
                             grad_scale, _ptr(adam_step_dev), _ptr(cursor_dev), _stream()))


def sgd_step(theta, grad, buf, lr, momentum, step, grad_scale=1.0, step_dev=None, cursor_dev=None):
    check(_lib.dmf_sgd_step(_ptr(theta), _ptr(grad), _ptr(buf), theta.numel(), lr, momentum, step, grad_scale, _ptr(step_dev),
                            _ptr(cursor_dev), _stream()))


def rmsprop_step(theta, grad, sq, lr, alpha, eps=1e-8, grad_scale=1.0, cursor_dev=None):
    check(_lib.dmf_rmsprop_step(_ptr(theta), _ptr(grad), _ptr(sq), theta.numel(), lr, alpha, eps, grad_scale, _ptr(cursor_dev),
                                _stream()))


def grad_reduce_adam(shape, B, ws, theta, m, v, grad, lr, b1, b2, eps, step, adam_step_dev=None, cursor_dev=None,
                     loss=None, loss_hist=None):
    check(_lib.dmf_grad_reduce_adam(C.byref(shape), B, _ptr(ws), _ptr(theta), _ptr(m), _ptr(v), _ptr(grad),
                                    lr, b1, b2, eps, step, _ptr(adam_step_dev), _ptr(cursor_dev), _ptr(loss),
                                    _ptr(loss_hist), _stream()))


def forward_ce(shape, inp, theta, pool_w, labels, logits, loss, pred=None):
    """Evaluation forward + per-patch cross-entropy in the same launch (dmf_forward_ce); raises DmfError where the shape has no
    such kernel (the caller then uses forward() + its own loss)."""
    check(_lib.dmf_forward_ce(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(labels), _ptr(logits), _ptr(loss),
                              _ptr(pred), _stream()))


def train_plan_steps(shape, inp, theta, pool_w, labels, loss_scale, logits, loss, ws, m, v, lr, b1, b2, eps, adam_step_dev, cursor_dev,
                     loss_hist, n_steps):
    """n_steps fused train steps on consecutive batches of a resident plan, enqueued by one C loop (dmf_train_plan_steps)."""
    check(_lib.dmf_train_plan_steps(C.byref(shape), C.byref(inp), _ptr(theta), _ptr(pool_w), _ptr(labels), loss_scale, _ptr(logits),
                                    _ptr(loss), _ptr(ws), _ptr(m), _ptr(v), lr, b1, b2, eps, _ptr(adam_step_dev), _ptr(cursor_dev),
                                    _ptr(loss_hist), n_steps, _stream()))


def grad_reduce_xgmi_adam(shape, B, ws, theta, m, v, comm, lr, b1, b2, eps, grad_scale, adam_step_dev, cursor_dev=None,
                          loss=None, loss_hist=None):
    check(_lib.dmf_grad_reduce_xgmi_adam(C.byref(shape), B, _ptr(ws), _ptr(theta), _ptr(m), _ptr(v), C.byref(comm), lr, b1,
                                         b2, eps, grad_scale, _ptr(adam_step_dev), _ptr(cursor_dev), _ptr(loss),
                                         _ptr(loss_hist), _stream()))


def hip_runtime_of_torch():
    """The libamdhip64 this process already has mapped (torch's own), found in /proc/self/maps — dlopen by bare name could
    bind a second copy of the runtime, whose stream and graph handles mean nothing to the first.  None if not found."""
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                path = line.split(None, 5)[-1].strip() if line.count('/') else ''
                if 'libamdhip64.so' in path:
                    return C.CDLL(path)
    except OSError:
        pass
    return None


def xgmi_sizes(capacity, world):
    d, f = C.c_int64(), C.c_int64()
    check(_lib.dmf_xgmi_sizes(capacity, world, C.byref(d), C.byref(f)))
    return d.value, f.value


def xgmi_alloc(nbytes):
    p = C.c_void_p()
    check(_lib.dmf_xgmi_alloc(nbytes, C.byref(p)))
    return p.value


def xgmi_free(ptr):
    check(_lib.dmf_xgmi_free(C.c_void_p(ptr)))


def xgmi_export(ptr):
    h = C.create_string_buffer(64)
    check(_lib.dmf_xgmi_export(C.c_void_p(ptr), h))
    return h.raw


def xgmi_open(handle):
    p = C.c_void_p()
    check(_lib.dmf_xgmi_open(C.create_string_buffer(handle, 64), C.byref(p)))
    return p.value


def xgmi_close(ptr):
    check(_lib.dmf_xgmi_close(C.c_void_p(ptr)))


def xgmi_status(comm):
    st = C.c_int32()
    check(_lib.dmf_xgmi_status(C.byref(comm), C.byref(st)))
    return st.value


def xgmi_allreduce(comm, buf, n, seq):
    _dev(buf, torch.float32, 'buf')
    check(_lib.dmf_xgmi_allreduce(C.byref(comm), _ptr(buf), n, seq, _stream()))


def qua_params(dqtl):
    """cfg['dqtl'] -> QuaParams (train/loss_function.py:21,32,66-68)."""
    return QuaParams(alpha=dqtl['alpha'], beta=dqtl['beta'], gamma=dqtl['gamma'], epsilon=dqtl['epsilon'], tao=dqtl['tao'])


def qua_loss(logits, bs, labels, params, loss=None, dlogits=None, grad_scale=1.0, cursor=None, loss_hist=None,
             scaler_state=None):
    _dev(logits, torch.float32, 'logits'); _dev(labels, torch.int32, 'labels')
    K = logits.shape[1]
    if logits.dim() != 2 or logits.shape[0] != 4 * bs:
        raise DmfError('qua_loss wants logits [4*bs, K]')
    if cursor is None and labels.numel() < bs:
        raise DmfError('qua_loss wants one label per sample')
    if dlogits is not None:
        _dev(dlogits, torch.float32, 'dlogits')
        if dlogits.shape != logits.shape:
            raise DmfError('dlogits must have the shape of logits')
    check(_lib.dmf_qua_loss_scaled(_ptr(logits), bs, K, _ptr(labels), _ptr(cursor), C.byref(params), grad_scale,
                                   _ptr(scaler_state), _ptr(loss), _ptr(loss_hist), _ptr(dlogits), _stream()))


def pair_argmax(logits, bs, pred):
    _dev(logits, torch.float32, 'logits'); _dev(pred, torch.int32, 'pred')
    if logits.dim() != 2 or logits.shape[0] < 2 * bs or pred.numel() < bs:
        raise DmfError('pair_argmax wants logits [>=2*bs, K] and pred [bs]')
    check(_lib.dmf_pair_argmax(_ptr(logits), bs, logits.shape[1], _ptr(pred), _stream()))


def band_mean_scene(scene):
    """[Hp, Wp, C] resident scene -> [Hp, Wp, 1] band mean."""
    _dev(scene, torch.float32, 'scene')
    out = torch.empty(scene.shape[0], scene.shape[1], 1, device=scene.device)
    check(_lib.dmf_band_mean(_ptr(scene), 0, 1, scene.shape[0] * scene.shape[1], scene.shape[2], _ptr(out), _stream()))
    return out


def band_mean_patches(a):
    """[B, C, P, P] patches -> [B, 1, P, P] band mean."""
    _dev(a, torch.float32, 'a')
    out = torch.empty(a.shape[0], 1, a.shape[2], a.shape[3], device=a.device)
    if a.shape[0]:
        check(_lib.dmf_band_mean(_ptr(a), 1, a.shape[0], a.shape[2] * a.shape[3], a.shape[1], _ptr(out), _stream()))
    return out


def confusion_accum(pred, target, K, matrix):
    check(_lib.dmf_confusion_accum(_ptr(pred), _ptr(target), pred.numel(), K, _ptr(matrix), _stream()))


def labelmap_write(pred, xy, W, label_map):
    check(_lib.dmf_labelmap_write(_ptr(pred), _ptr(xy), pred.numel(), W, _ptr(label_map), _stream()))


def pan2ms(pan, H, W, out):
    check(_lib.dmf_pan2ms(_ptr(pan), pan.shape[1], H, W, _ptr(out), _stream()))
