"""Condition encoders of the conditional pipelines.  `SparseRangeImageEncoder2` (ldm/encoders.py:86-95) is a pure
re-indexing -- out[b, (w%4)*C + c, w//4, h] = in[b, c, w, h] -- so it is a view/permute on the device tensor;
the concat with the latents happens inside the conv_in input packing kernel."""
import torch


class SparseRangeImageEncoder2:
    def encode(self, x):
        return self(x)

    def __call__(self, x):
        B, C, W, H = x.shape
        if W % 4:
            raise ValueError("azimuth extent must be a multiple of 4")
        x = torch.flatten(x.permute(0, 2, 1, 3), start_dim=1, end_dim=2)
        return x.reshape(B, W // 4, C * 4, H).permute(0, 2, 1, 3)
