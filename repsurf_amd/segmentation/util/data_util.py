"""util.data_util — the packed-batch wire format of the reference (segmentation/util/data_util.py:15-23)."""
import torch


def collate_fn(batch):
    """[(coord (n_i,3), feat (n_i,C), label (n_i,)|None), ...] -> (coord, feat, label|None, offset (B,) int32 running
    row ends).  The host copy of the offsets stays attached to the tensor (and follows `.to(device)` through
    repsurf_amd.ops.offsets_tensor), so the modules never read them back from the device."""
    coord, feat, label = list(zip(*batch))
    offset, count = [], 0
    for item in coord:
        count += item.shape[0]
        offset.append(count)
    off = torch.IntTensor(offset)
    off._rs_host = ((off.data_ptr(), off._version), tuple(offset))      # what repsurf_amd.ops.host_offsets would read
    return torch.cat(coord), torch.cat(feat), torch.cat(label) if label[0] is not None else None, off
