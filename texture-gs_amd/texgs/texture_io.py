"""Texture / checkpoint I/O of the hot path's data formats: what sits either side of the operator
in retexture.py / extract_texture.py, so that a reference checkpoint or a cubemap-cross PNG can be fed to (or produced
from) `GaussianRasterizer` without the reference's model classes (which need tinycudann / cv2 / nvdiffrast).

* texel <-> RGB: texels are SH-DC coefficients, rgb = C0 * t + 0.5 clamped (models/texture_gaussian3d.py:16-21)
* cubemap cross (3R x 4R x 3, faces at the positions of models/texture_gaussian3d.py:451-474) <-> texture[6,R,R,3]
* change_texture modes -1..3 (models/texture_gaussian3d.py:463-495) -- what retexture.py:58 applies to a loaded PNG
* PNG load with the resize / channel handling of retexture.py:48-57, PNG save as extract_texture.py:15-17
* the checkpoint tuple `(state_dict, iteration)` of train.py:181-184 with the state_dict layout of
  models/texture_gaussian3d.py:145-172: parameter tuple (_xyz, _scaling, _rotation, _opacity, _shs, _texture) holding RAW
  values (log-scale, un-normalised quaternion, logit opacity) and their activations (:196-240).

Pure host code (torch / numpy / PIL): byte and layout work around the device path, no kernels.  Pinned against the
reference's own functions by tests/golden/texture_io.npz (tests/golden/make_golden.py runs rgb2sh0 / sh02rgb / cube_map /
change_texture extracted from the reference source).
"""
from typing import NamedTuple, Optional

import numpy as np
import torch

SH_C0 = 0.28209479177387814


def rgb2sh0(rgb: torch.Tensor) -> torch.Tensor:
    return (rgb - 0.5) / SH_C0


def sh02rgb(sh0: torch.Tensor) -> torch.Tensor:
    return torch.clamp(SH_C0 * sh0 + 0.5, 0.0, 1.0)


# (row block, column block) of each cube face inside the 3 x 4 cross: +x, -x, +y, -y, +z, -z
_CROSS_POS = ((1, 2), (1, 0), (0, 1), (2, 1), (1, 1), (1, 3))


def cube_to_cross(rgb_faces: torch.Tensor) -> torch.Tensor:
    """[6,R,R,3] RGB faces -> [3R,4R,3] cross image (unused cells zero)."""
    if rgb_faces.dim() != 4 or rgb_faces.shape[0] != 6 or rgb_faces.shape[1] != rgb_faces.shape[2] or rgb_faces.shape[3] != 3:
        raise ValueError(f"expected [6,R,R,3], got {tuple(rgb_faces.shape)}")
    R = rgb_faces.shape[1]
    cross = torch.zeros(3 * R, 4 * R, 3, dtype=rgb_faces.dtype, device=rgb_faces.device)
    for f, (rb, cb) in enumerate(_CROSS_POS):
        cross[rb * R:(rb + 1) * R, cb * R:(cb + 1) * R] = rgb_faces[f]
    return cross


def cross_to_cube(cross: torch.Tensor) -> torch.Tensor:
    """[3r,4r,3] cross image -> [6,r,r,3] faces."""
    r = cross.shape[0] // 3
    if tuple(cross.shape) != (3 * r, 4 * r, 3):
        raise ValueError(f"a cubemap cross must be [3r,4r,3], got {tuple(cross.shape)}")
    return torch.stack([cross[rb * r:(rb + 1) * r, cb * r:(cb + 1) * r] for rb, cb in _CROSS_POS], dim=0)


def texture_to_cross(texture_sh0: torch.Tensor) -> torch.Tensor:
    """The cross image extract_texture.py writes: clamped RGB of the SH-DC texture."""
    return cube_to_cross(sh02rgb(texture_sh0))


def change_texture(texture_sh0: torch.Tensor, cross_rgb: torch.Tensor, mode: int = 0) -> torch.Tensor:
    """New SH-DC texture from a cross image (same resolution as the texture) and the current texture.
    mode -1 replace; 0 modulate by the clamped 3x luminance of the old texture; 1 multiply; 2 divide old by new;
    3 tint the non-black texels of the new texture with twice the old luminance, then add."""
    new = cross_to_cube(cross_rgb)
    old = sh02rgb(texture_sh0)
    if new.shape != old.shape:
        raise ValueError(f"cross image resolves to {tuple(new.shape)}, texture is {tuple(old.shape)}")
    if mode == -1:
        pass
    elif mode == 0:
        new = new * (old * 3).clamp(0, 1).mean(dim=-1, keepdim=True)
    elif mode == 1:
        new = new * old
    elif mode == 2:
        new = old / new
    elif mode == 3:
        mask = new.sum(-1) > 0.01
        tint = old.clone()
        tint[mask] = 2 * old[mask].mean(-1)[..., None] * new[mask]
        new = new + tint
    else:
        raise ValueError("mode must be -1, 0, 1, 2 or 3")
    return rgb2sh0(new)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [h,w,c] -> uint8 [out_h,out_w,c], 2-tap bilinear with half-pixel centres and edge clamp (the geometry of
    cv2.resize(..., interpolation=cv2.INTER_LINEAR), retexture.py:54; cv2 interpolates in 11-bit fixed point, so single
    values may differ from it by one grey level)."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()

    def taps(n_in, n_out):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(src).astype(np.int64)
        f = src - i0
        i1 = np.clip(i0 + 1, 0, n_in - 1)
        f = np.where(i0 < 0, 0.0, f)
        i0 = np.clip(i0, 0, n_in - 1)
        return i0, i1, f
    y0, y1, fy = taps(h, out_h)
    x0, x1, fx = taps(w, out_w)
    a = img.astype(np.float64)
    top = a[y0][:, x0] * (1 - fx)[None, :, None] + a[y0][:, x1] * fx[None, :, None]
    bot = a[y1][:, x0] * (1 - fx)[None, :, None] + a[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def load_cross_png(path: str, resolution: int, device=None) -> torch.Tensor:
    """PNG cross image -> float32 RGB [3R,4R,3] in [0,1], resized to the texture's resolution (retexture.py:48-57)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"))
    r = img.shape[0] // 3
    if img.shape != (3 * r, 4 * r, 3):
        raise ValueError(f"{path}: a cubemap cross must be 3r x 4r pixels, got {img.shape[1]} x {img.shape[0]}")
    img = resize_bilinear_u8(img, 3 * resolution, 4 * resolution)
    return torch.tensor(img.astype(np.float32) / 255.0, device=device)


def save_cross_png(texture_sh0: torch.Tensor, path: str) -> None:
    """SH-DC texture -> cross PNG, values truncated to uint8 like extract_texture.py:15-17."""
    from PIL import Image
    cross = (torch.clamp(texture_to_cross(texture_sh0), 0, 1) * 255).detach().cpu().numpy().astype(np.uint8)
    Image.fromarray(cross, "RGB").save(path)


class GaussianState(NamedTuple):
    """Stage-3 parameters as a checkpoint holds them (RAW values) plus what is needed to call the operator."""
    xyz: torch.Tensor            # [N,3]
    scaling: torch.Tensor        # [N,3] log-scales
    rotation: torch.Tensor       # [N,4] un-normalised (w,x,y,z)
    opacity: torch.Tensor        # [N,1] logits
    shs: torch.Tensor            # [N,K,3] view-dependent SH coefficients 1..K
    texture: torch.Tensor        # [6,R,R,3] SH-DC texels
    active_sh_degree: int
    spatial_lr_scale: float
    net_state: Optional[tuple]   # (uv_net, inv_uv_net, geo_emb) state dicts, passed through
    optim_state: Optional[tuple]
    iteration: int

    # activations of models/texture_gaussian3d.py:196-240
    def get_scaling(self):
        return torch.exp(self.scaling)

    def get_rotation(self):
        return torch.nn.functional.normalize(self.rotation)

    def get_opacity(self):
        return torch.sigmoid(self.opacity)


def _plain(t):
    return t.detach() if isinstance(t, torch.Tensor) else t


def load_checkpoint(path: str, map_location="cpu", weights_only: bool = True) -> GaussianState:
    """`torch.save((model.state_dict(), iteration), path)` of train.py:181-184 for the TextureGaussian3D model."""
    obj = torch.load(path, map_location=map_location, weights_only=weights_only)
    if not (isinstance(obj, (tuple, list)) and len(obj) == 2 and isinstance(obj[0], dict)):
        raise ValueError("not a Texture-GS checkpoint: expected the tuple (state_dict, iteration)")
    sd, iteration = obj
    for key in ("hyperparams", "params"):
        if key not in sd:
            raise ValueError(f"checkpoint state_dict has no '{key}' entry")
    if len(sd["params"]) != 6:
        raise ValueError("expected the stage-3 parameter tuple (_xyz, _scaling, _rotation, _opacity, _shs, _texture); "
                         f"got {len(sd['params'])} entries (a stage-1/2 checkpoint?)")
    xyz, scaling, rotation, opacity, shs, texture = (_plain(t) for t in sd["params"])
    deg, lr_scale = sd["hyperparams"]
    return GaussianState(xyz, scaling, rotation, opacity, shs, texture, int(deg), float(lr_scale), sd.get("net_state"),
                         sd.get("optim_state"), int(iteration))


def save_checkpoint(state: GaussianState, path: str) -> None:
    sd = dict(hyperparams=(state.active_sh_degree, state.spatial_lr_scale),
              optim_state=state.optim_state if state.optim_state is not None else (),
              net_state=state.net_state if state.net_state is not None else (),
              params=(state.xyz, state.scaling, state.rotation, state.opacity, state.shs, state.texture))
    torch.save((sd, state.iteration), path)
