"""`slice_acquisition` / `slice_acquisition_adjoint` with the reference's signatures
(svort/slice_acquisition/slice_acq.py:193-263), served by the gfx950 kernels of fsg_slice_acq.hip.

Tensor layouts are the reference's: vol (1,1,D,H,W), slices (n,1,h,w), masks bool of the same shapes,
transforms (n,3,4), psf (pd,ph,pw).  `semantics` selects which of the reference's two arithmetics is
followed: "cuda" (default; what the reference computes on a GPU) or "torch" (its CPU fallback, for
seed-matched comparison with a CPU run of the reference).  There is no CPU path here.
"""
from __future__ import annotations

from .... import kernels as K

_SEMANTICS = "cuda"


def set_semantics(name: str) -> str:
    """Process-wide default for `semantics`; returns the previous value."""
    global _SEMANTICS
    if name not in ("cuda", "torch"):
        raise ValueError("semantics must be 'cuda' or 'torch'")
    prev, _SEMANTICS = _SEMANTICS, name
    return prev


def get_semantics() -> str:
    return _SEMANTICS


def _dev(t, device):
    return None if t is None else t.to(device).contiguous()


def slice_acquisition(transforms, vol, vol_mask, slices_mask, psf, slice_shape, res_slice, need_weight, interp_psf,
                      semantics=None):
    sem = semantics or _SEMANTICS
    dev = vol.device
    v = vol.reshape(vol.shape[-3:]).contiguous()
    n, (h, w) = transforms.shape[0], slice_shape
    vm = None if vol_mask is None or vol_mask.numel() == 0 else _dev(vol_mask, dev).reshape(v.shape)
    sm = None if slices_mask is None or slices_mask.numel() == 0 else _dev(slices_mask, dev).reshape(n, h, w)
    out = K.slice_acq_forward(_dev(transforms.float(), dev), v, vm, sm, _dev(psf.float(), dev), (h, w), res_slice,
                              need_weight=need_weight, interp_psf=interp_psf, semantics=sem)
    if need_weight:
        return out[0].view(n, 1, h, w), out[1].view(n, 1, h, w)
    return out.view(n, 1, h, w)


def slice_acquisition_adjoint(transforms, psf, slices, slices_mask, vol_mask, vol_shape, res_slice, interp_psf,
                              equalize, semantics=None, slice_ids=None):
    sem = semantics or _SEMANTICS
    dev = slices.device
    h, w = slices.shape[-2:]
    s = slices.reshape(-1, h, w).contiguous()
    D, H, W = (int(v) for v in vol_shape)
    n = transforms.shape[0]
    vm = None if vol_mask is None or vol_mask.numel() == 0 else _dev(vol_mask, dev).reshape(D, H, W)
    sm = None if slices_mask is None or slices_mask.numel() == 0 else _dev(slices_mask, dev).reshape(n, h, w)
    vol = K.slice_acq_adjoint(_dev(transforms.float(), dev), _dev(psf.float(), dev), s, sm, vm, (D, H, W), res_slice,
                              interp_psf=interp_psf, equalize=equalize, semantics=sem, slice_ids=slice_ids)
    return vol.view(1, 1, D, H, W)
