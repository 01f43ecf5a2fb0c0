"""TEST INFRASTRUCTURE ONLY (never imported by flair_amd/ or the timed part of bench.py).

Literal numpy restatement of the modulated deformable convolution (DCNv2) forward that ships as SOURCE in the
reference: guided_diffusion/dcn/src/deform_conv_cuda_kernel.cu and deform_conv_cuda.cpp.  It is the same operator
the hot path calls as torchvision.ops.deform_conv2d (unet_new.py:889-898), so it pins the semantics of
oracle/thirdparty.deform_conv2d -- channel order of the offsets, the `> -1 && < size` sampling window, the
per-corner validity rules -- to text the reference itself holds, independently of F.grid_sample.

Every function names the lines it follows.  Loops follow the CUDA kernel (one "thread" per (c_im, b, h_col, w_col),
kernel_h x kernel_w taps inside); only the (h_col, w_col) plane is vectorised with numpy, element for element.
Arithmetic is done in the dtype of `x` (float32 like the reference's scalar_t = float, or float64).
"""
import numpy as np


def dmcn_im2col_bilinear(bottom_data, data_width, height, width, h, w):
    """deform_conv_cuda_kernel.cu:468-497.  bottom_data: one (height, data_width) channel plane; h, w: arrays of
    sampling positions (already inside the (-1, size) window).  Returns the sampled values."""
    h_low = np.floor(h).astype(np.int64)                      # :472
    w_low = np.floor(w).astype(np.int64)                      # :473
    h_high = h_low + 1                                        # :474
    w_high = w_low + 1                                        # :475
    lh = h - h_low.astype(h.dtype)                            # :477
    lw = w - w_low.astype(w.dtype)                            # :478
    hh, hw = 1 - lh, 1 - lw                                   # :479
    flat = bottom_data.reshape(-1)

    def at(ok, hi, wi):                                       # :481-492: value where the condition holds, else 0
        idx = np.where(ok, hi * data_width + wi, 0)
        return np.where(ok, flat[idx], np.zeros((), dtype=bottom_data.dtype))

    v1 = at((h_low >= 0) & (w_low >= 0), h_low, w_low)                            # :481-483
    v2 = at((h_low >= 0) & (w_high <= width - 1), h_low, w_high)                  # :484-486
    v3 = at((h_high <= height - 1) & (w_low >= 0), h_high, w_low)                 # :487-489
    v4 = at((h_high <= height - 1) & (w_high <= width - 1), h_high, w_high)       # :490-492
    w1, w2, w3, w4 = hh * hw, hh * lw, lh * hw, lh * lw                           # :494
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4                                  # :496


def modulated_deformable_im2col(data_im, data_offset, data_mask, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w,
                                dilation_h, dilation_w, deformable_group, height_col, width_col):
    """deform_conv_cuda_kernel.cu:571-633 (launched by :760-783 with batch_size = 1 per image, as
    deform_conv_cuda.cpp:541-545 does).  data_im (C, H, W); data_offset (G*2*kh*kw, Hc, Wc); data_mask (G*kh*kw, Hc, Wc).
    Returns columns (C*kh*kw, Hc*Wc)."""
    channels, height, width = data_im.shape
    channel_per_deformable_group = channels // deformable_group                   # :770
    dt = data_im.dtype
    data_col = np.zeros((channels * kernel_h * kernel_w, height_col, width_col), dtype=dt)
    h_col = np.arange(height_col).reshape(-1, 1)
    w_col = np.arange(width_col).reshape(1, -1)
    for c_im in range(channels):                                                  # :588 (one thread per c_im, h_col, w_col)
        c_col = c_im * kernel_h * kernel_w                                        # :589
        deformable_group_index = c_im // channel_per_deformable_group             # :592
        h_in = h_col * stride_h - pad_h                                           # :594
        w_in = w_col * stride_w - pad_w                                           # :595
        data_im_ptr = data_im[c_im]                                               # :599
        off_base = deformable_group_index * 2 * kernel_h * kernel_w               # :600
        mask_base = deformable_group_index * kernel_h * kernel_w                  # :602
        for i in range(kernel_h):                                                 # :604
            for j in range(kernel_w):                                             # :606
                offset_h = data_offset[off_base + 2 * (i * kernel_w + j)]         # :608, :611
                offset_w = data_offset[off_base + 2 * (i * kernel_w + j) + 1]     # :609, :612
                mask = data_mask[mask_base + i * kernel_w + j]                    # :610, :613
                h_im = (h_in + i * dilation_h).astype(dt) + offset_h              # :615
                w_im = (w_in + j * dilation_w).astype(dt) + offset_w              # :616
                inside = (h_im > -1) & (w_im > -1) & (h_im < height) & (w_im < width)   # :618
                hs = np.where(inside, h_im, np.zeros((), dtype=dt))
                ws = np.where(inside, w_im, np.zeros((), dtype=dt))
                val = np.where(inside, dmcn_im2col_bilinear(data_im_ptr, width, height, width, hs, ws),
                               np.zeros((), dtype=dt))                            # :614, :625
                data_col[c_col + i * kernel_w + j] = val * mask                   # :627-628
    return data_col.reshape(channels * kernel_h * kernel_w, height_col * width_col)


def modulated_deform_conv_forward(x, weight, bias, offset, mask, stride=(1, 1), padding=(1, 1), dilation=(1, 1),
                                  group=1, deformable_group=1):
    """deform_conv_cuda.cpp:490-567: per image im2col (:541-545), per weight group addmm of weight[g].flatten(1) with
    columns[g] (:552-557), bias added at the end (:564-566).  x (N, C, H, W); weight (Cout, C/group, kh, kw);
    offset (N, 2*G*kh*kw, Hout, Wout); mask (N, G*kh*kw, Hout, Wout)."""
    x = np.asarray(x)
    dt = x.dtype
    weight, offset, mask = np.asarray(weight, dtype=dt), np.asarray(offset, dtype=dt), np.asarray(mask, dtype=dt)
    batch, channels, height, width = x.shape                                      # :501-504
    channels_out, channels_kernel, kernel_h, kernel_w = weight.shape              # :506-509
    assert channels == channels_kernel * group                                    # :514-516
    height_out = (height + 2 * padding[0] - (dilation[0] * (kernel_h - 1) + 1)) // stride[0] + 1   # :518-519
    width_out = (width + 2 * padding[1] - (dilation[1] * (kernel_w - 1) + 1)) // stride[1] + 1     # :520-521
    out = np.zeros((batch, channels_out, height_out, width_out), dtype=dt)        # :530
    for b in range(batch):                                                        # :540
        columns = modulated_deformable_im2col(x[b], offset[b], mask[b], kernel_h, kernel_w, padding[0], padding[1],
                                              stride[0], stride[1], dilation[0], dilation[1], deformable_group,
                                              height_out, width_out)              # :541-545
        wg = weight.reshape(group, channels_out // group, -1)                     # :548-549, flatten(1) of :555
        cg = columns.reshape(group, columns.shape[0] // group, columns.shape[1])  # :550
        for g in range(group):                                                    # :552
            out[b].reshape(group, channels_out // group, -1)[g] += wg[g] @ cg[g]  # :553-556
    if bias is not None:                                                          # :564
        out += np.asarray(bias, dtype=dt).reshape(1, -1, 1, 1)                    # :565
    return out
