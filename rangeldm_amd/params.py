"""State-dict inventories (diffusers layout) for the UNet and VAE of the RangeLDM hot path.

Key names follow what `UNet2DModel` / `AutoencoderKL` register after the reference's surgery
(ldm/utils.py:98-105 -- `Downsample2D(name="op")` registers only `conv`), so a checkpoint written by
ldm/train_unconditional.py:654-675 (`pipeline.save_pretrained`) maps 1:1 onto `load_state_dict`.

`sgm_to_diffusers_vae_key` restates the key renaming of ldm/convert_vae.py:25-120 (+ diffusers'
`renew_vae_resnet_paths`): sgm `Encoder`/`Decoder` checkpoints -> diffusers `AutoencoderKL` names.
"""
from collections import OrderedDict
from .config import UNetConfig, VAEConfig


def _conv(d, name, cout, cin, k):
    d[name + ".weight"] = (cout, cin, k, k)
    d[name + ".bias"] = (cout,)


def _lin(d, name, cout, cin):
    d[name + ".weight"] = (cout, cin)
    d[name + ".bias"] = (cout,)


def _norm(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)


def _resnet(d, p, cin, cout, temb):
    _norm(d, p + ".norm1", cin)
    _conv(d, p + ".conv1", cout, cin, 3)
    if temb:
        _lin(d, p + ".time_emb_proj", cout, temb)
    _norm(d, p + ".norm2", cout)
    _conv(d, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, p + ".conv_shortcut", cout, cin, 1)


def _attn(d, p, c):
    _norm(d, p + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v"):
        _lin(d, p + "." + n, c, c)
    _lin(d, p + ".to_out.0", c, c)


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    """(name -> shape) in module-registration order.  RangeLDM cfg totals 30 135 684 parameters (= README 115 MB)."""
    d = OrderedDict()
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    _conv(d, "conv_in", boc[0], cfg.in_channels, 3)
    _lin(d, "time_embedding.linear_1", temb, boc[0])
    _lin(d, "time_embedding.linear_2", temb, temb)
    out = boc[0]
    for i, btype in enumerate(cfg.down_block_types):
        cin, out = out, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(d, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb)
        if btype == "AttnDownBlock2D":
            for j in range(cfg.layers_per_block):
                _attn(d, f"down_blocks.{i}.attentions.{j}", out)
        if i != len(boc) - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    c = boc[-1]
    _resnet(d, "mid_block.resnets.0", c, c, temb)
    if cfg.add_attention:
        _attn(d, "mid_block.attentions.0", c)
    _resnet(d, "mid_block.resnets.1", c, c, temb)
    rev = tuple(reversed(boc))
    out = rev[0]
    for i, btype in enumerate(cfg.up_block_types):
        prev, out = out, rev[i]
        inp = rev[min(i + 1, len(boc) - 1)]
        n = cfg.layers_per_block + 1
        for j in range(n):
            skip = inp if j == n - 1 else out
            rin = prev if j == 0 else out
            _resnet(d, f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb)
        if btype == "AttnUpBlock2D":
            for j in range(n):
                _attn(d, f"up_blocks.{i}.attentions.{j}", out)
        if i != len(boc) - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    _norm(d, "conv_norm_out", boc[0])
    _conv(d, "conv_out", cfg.out_channels, boc[0], 3)
    return d


def vae_param_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    """diffusers AutoencoderKL keys for the attention-free, quant-conv-free VAE the reference builds
    (ldm/convert_vae.py:149-189: quant convs -> Identity, attention -> identity)."""
    d = OrderedDict()
    L = len(cfg.ch_mult)
    chs = [cfg.ch * m for m in cfg.ch_mult]
    # encoder (vae/sgm/modules/diffusionmodules/model.py:707-896)
    _conv(d, "encoder.conv_in", cfg.ch, cfg.in_channels, 3)
    cin = cfg.ch
    for i in range(L):
        for j in range(cfg.num_res_blocks):
            _resnet(d, f"encoder.down_blocks.{i}.resnets.{j}", cin, chs[i], 0)
            cin = chs[i]
        if i != L - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", cin, cin, 3)
    _resnet(d, "encoder.mid_block.resnets.0", cin, cin, 0)
    _resnet(d, "encoder.mid_block.resnets.1", cin, cin, 0)
    _norm(d, "encoder.conv_norm_out", cin)
    _conv(d, "encoder.conv_out", 2 * cfg.z_channels if cfg.double_z else cfg.z_channels, cin, 3)
    # decoder (model.py:899-1057); diffusers up_blocks.{i} == sgm up.{L-1-i}
    cin = chs[-1]
    _conv(d, "decoder.conv_in", cin, cfg.z_channels, 3)
    _resnet(d, "decoder.mid_block.resnets.0", cin, cin, 0)
    _resnet(d, "decoder.mid_block.resnets.1", cin, cin, 0)
    for i in range(L):
        cout = chs[L - 1 - i]
        for j in range(cfg.num_res_blocks + 1):
            _resnet(d, f"decoder.up_blocks.{i}.resnets.{j}", cin, cout, 0)
            cin = cout
        if i != L - 1:
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", cin, cin, 3)
    _norm(d, "decoder.conv_norm_out", cin)
    _conv(d, "decoder.conv_out", cfg.out_channels, cin, 3)
    return d


def sgm_to_diffusers_vae_key(key: str, num_levels: int = 3):
    """Map one sgm `AutoencodingEngine` state-dict key to the diffusers AutoencoderKL key.
    Restates ldm/convert_vae.py:25-120.  Returns None for keys the hot path drops (loss / discriminator)."""
    if not (key.startswith("encoder.") or key.startswith("decoder.")):
        return None
    side, rest = key.split(".", 1)
    parts = rest.split(".")
    rename_leaf = {"nin_shortcut": "conv_shortcut", "norm_out": "conv_norm_out"}
    if parts[0] in ("conv_in", "conv_out", "norm_out"):
        parts[0] = rename_leaf.get(parts[0], parts[0])
        return side + "." + ".".join(parts)
    if parts[0] == "mid":
        blk = {"block_1": "0", "block_2": "1"}.get(parts[1])
        if blk is None:
            return None                                   # mid.attn_1: VAE is attention-free (attn_type: none)
        tail = [rename_leaf.get(p, p) for p in parts[2:]]
        return f"{side}.mid_block.resnets.{blk}." + ".".join(tail)
    if parts[0] in ("down", "up"):
        lvl = int(parts[1])
        if parts[0] == "up":
            lvl = num_levels - 1 - lvl
        blocks = "down_blocks" if parts[0] == "down" else "up_blocks"
        if parts[2] == "block":
            tail = [rename_leaf.get(p, p) for p in parts[4:]]
            return f"{side}.{blocks}.{lvl}.resnets.{parts[3]}." + ".".join(tail)
        if parts[2] == "downsample":
            return f"{side}.{blocks}.{lvl}.downsamplers.0." + ".".join(parts[3:])
        if parts[2] == "upsample":
            return f"{side}.{blocks}.{lvl}.upsamplers.0." + ".".join(parts[3:])
    return None


def count_params(shapes) -> int:
    n = 0
    for s in shapes.values():
        k = 1
        for v in s:
            k *= v
        n += k
    return n
