"""CPU restatement of GanMonitor.stitch_subvolumes (custom_callback.py:47-223), 3-D branch -- TEST INFRASTRUCTURE ONLY.
Follows the reference statement by statement (numpy), with `gen` any callable mapping a [1,kX,kY,kZ,1] float array to the
same shape.  The TIFF writing at :204-223 is I/O and not restated.  PARITY UNPINNED (the reference module cannot be
imported: it imports tensorflow/skimage at the top)."""
import numpy as np


def min_max_norm(data):
    """utils.py:11-24."""
    dmin, dmax = np.min(data), np.max(data)
    if (dmax - dmin) == 0:
        raise ValueError("Cannot perform min-max normalization when max and min are equal.")
    return (data - dmin) / (dmax - dmin)


def process_imaging_otf(arr):
    """main.py:169-177 with axis=None, keepdims=False (as called at custom_callback.py:172)."""
    mx, mn = arr.max(), arr.min()
    return 2.0 * (arr - mn) / (mx - mn) - 1.0


def stitch_subvolumes(gen, img, subvol_size, stride=(25, 25, 128), complete=False, padFactor=0.25, border_removal=True,
                      process_img=False):
    """img: [X,Y,Z,C]; subvol_size: (batch, kH, kW, kD, C) as in the reference (custom_callback.py:108)."""
    if complete:                                                                 # :82-104
        xspacing, yspacing = int(padFactor * img.shape[0]), int(padFactor * img.shape[1])
        oimgshape = img.shape
        if stride[2] == 1:
            img = np.pad(img, ((xspacing, xspacing), (yspacing, yspacing), (0, 0), (0, 0)), 'symmetric')
        else:
            zspacing = int(padFactor * img.shape[2])
            img = np.pad(img, ((xspacing, xspacing), (yspacing, yspacing), (zspacing, zspacing), (0, 0)), 'symmetric')
    H, W, D, C = img.shape                                                       # :106-109
    kH, kW, kD = subvol_size[1], subvol_size[2], subvol_size[3]
    if not complete or not border_removal:                                       # :111-116
        pH, pW, pD = 0, 0, 0
    else:
        pH, pW, pD = int(0.1 * kH), int(0.1 * kW), int(0.1 * kD)
        if kD == D:
            pD = 0
    pix_tracker = np.zeros([H, W, D, C], dtype='float32')
    pred = np.zeros(img.shape, dtype='float32')
    sh, sw, sd = stride
    dim_out_h = int(np.floor((H - kH) / sh + 1))                                 # :126-128
    dim_out_w = int(np.floor((W - kW) / sw + 1))
    dim_out_d = int(np.floor((D - kD) / sd + 1))
    start_row = 0
    for i in range(dim_out_h + 1):                                               # :140-190
        start_col = 0
        if start_row > H - kH:
            start_row = H - kH
        for j in range(dim_out_w + 1):
            start_dep = 0
            if start_col > W - kW:
                start_col = W - kW
            for k in range(dim_out_d + 1):
                if start_dep > D - kD:
                    start_dep = D - kD
                pix_tracker[start_row + pH:(start_row + kH - pH), start_col + pW:(start_col + kW - pW),
                            start_dep + pD:(start_dep + kD - pD)] += 1.
                arr = img[start_row:(start_row + kH), start_col:(start_col + kW), start_dep:(start_dep + kD)]
                if process_img:
                    arr = process_imaging_otf(arr)
                arr = gen(np.expand_dims(arr, axis=0))[0]
                arr = arr[pH:kH - pH, pW:kW - pW, pD:kD - pD]
                pred[start_row + pH:(start_row + kH - pH), start_col + pW:(start_col + kW - pW),
                     start_dep + pD:(start_dep + kD - pD)] += arr
                start_dep += sd
            start_col += sw
        start_row += sh
    with np.errstate(invalid='ignore', divide='ignore'):
        pred = np.true_divide(pred, pix_tracker)                                 # :192
    if complete:                                                                 # :195-200
        if stride[2] == 1:
            pred = pred[xspacing:oimgshape[0] + xspacing, yspacing:oimgshape[1] + yspacing, ]
        else:
            pred = pred[xspacing:oimgshape[0] + xspacing, yspacing:oimgshape[1] + yspacing,
                        zspacing:oimgshape[2] + zspacing, ]
    return 255 * min_max_norm(pred)                                              # :202
