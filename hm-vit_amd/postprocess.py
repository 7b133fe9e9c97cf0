"""Detection post-processing backed by libhmvit (HIP, gfx950): mirror of the reference's
``VoxelPostprocessor`` (``opencood/data_utils/post_processor/voxel_postprocessor.py:18-396``) for the inference
side - ``generate_anchor_box`` and ``post_process`` - and of ``opencood/utils/eval_utils.py`` (``caluclate_tp_fp``,
``calculate_ap``, ``voc_ap``).  Same constructor dict (the yaml's ``postprocess`` block), same call signatures and
return values (``(pred_box3d_tensor (N, 8, 3), scores (N))`` or ``(None, None)``).

Box decoding, the two sanity filters, the score ranking, the polygon IoU and the greedy rotated NMS all run on the
device (csrc/post.hip); the reference does them in numpy + shapely on the host.  The AP bookkeeping (sorting a frame's
detections, greedy GT matching, VOC-2010 interpolation) is the reference's own pure-Python arithmetic on a few hundred
numbers and stays on the host, with the IoU matrix coming from the device.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import _lib

GT_RANGE = [-102.4, -102.4, -3, 102.4, 102.4, 1]   # opencood/data_utils/datasets/__init__.py:24


def boxes_to_corners_3d(boxes: np.ndarray) -> np.ndarray:
    """(n, 7) [x, y, z, h, w, l, yaw] ('hwl' order) -> (n, 8, 3) corners (box_utils.py:143-190)."""
    b = np.asarray(boxes).astype(np.float32).copy()
    b[:, 3:6] = b[:, [5, 4, 3]]
    template = np.array([[1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1], [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1]],
                        dtype=np.float32) / 2
    c = b[:, None, 3:6] * template[None]
    cosa, sina = np.cos(b[:, 6]), np.sin(b[:, 6])
    rot = np.zeros((len(b), 3, 3), dtype=np.float32)
    rot[:, 0, 0] = cosa; rot[:, 0, 1] = sina; rot[:, 1, 0] = -sina; rot[:, 1, 1] = cosa; rot[:, 2, 2] = 1
    return np.einsum("nkc,ncd->nkd", c, rot) + b[:, None, 0:3]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("hm-vit_amd post-processing runs on the GPU only: pass CUDA tensors")
    return t.contiguous().float()


def quad_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """IoU (Na, Nb) of the convex quadrilaterals given by the first four corners (x, y) of ``a`` (Na, 8, 3) or
    (Na, 4, 2) and ``b`` (common_utils.compute_iou / convert_format, :120-158)."""
    a, b = _dev_f32(a), _dev_f32(b)
    na, nb = a.shape[0], b.shape[0]
    out = torch.empty(na, nb, device=a.device, dtype=torch.float32)
    if na and nb:
        if a.shape[1:] != b.shape[1:]:
            raise ValueError("quad_iou: both box sets must use the same corner layout")
        stride_pt = a.shape[2]
        stride_box = a.shape[1] * a.shape[2]
        _lib.check(_lib.lib.hmvit_quad_iou(a.data_ptr(), b.data_ptr(), na, nb, stride_box, stride_pt, out.data_ptr(), _stream()),
                   "quad_iou")
    return out


class VoxelPostprocessor:
    def __init__(self, anchor_params: dict, train: bool = False):
        self.params = anchor_params
        self.bbx_dict = {}
        self.train = train
        self.anchor_num = self.params["anchor_args"]["num"]

    # voxel_postprocessor.py:24-72 (host-side table, identical arithmetic)
    def generate_anchor_box(self) -> np.ndarray:
        a = self.params["anchor_args"]
        W, H, l, w, h = a["W"], a["H"], a["l"], a["w"], a["h"]
        r = [math.radians(e) for e in a["r"]]
        assert self.anchor_num == len(r)
        vh, vw = a["vh"], a["vw"]
        xrange = [a["cav_lidar_range"][0], a["cav_lidar_range"][3]]
        yrange = [a["cav_lidar_range"][1], a["cav_lidar_range"][4]]
        stride = a.get("feature_stride", 2)
        x = np.linspace(xrange[0] + vw, xrange[1] - vw, W // stride)
        y = np.linspace(yrange[0] + vh, yrange[1] - vh, H // stride)
        cx, cy = np.meshgrid(x, y)
        cx = np.tile(cx[..., np.newaxis], self.anchor_num)
        cy = np.tile(cy[..., np.newaxis], self.anchor_num)
        cz = np.ones_like(cx) * -1.0
        w_, l_, h_ = np.ones_like(cx) * w, np.ones_like(cx) * l, np.ones_like(cx) * h
        r_ = np.ones_like(cx)
        for i in range(self.anchor_num):
            r_[..., i] = r[i]
        if self.params["order"] == "hwl":
            return np.stack([cx, cy, cz, h_, w_, l_, r_], axis=-1)
        if self.params["order"] == "lhw":
            return np.stack([cx, cy, cz, l_, h_, w_, r_], axis=-1)
        raise ValueError("Unknown bbx order.")

    # voxel_postprocessor.py:74-194 (training targets; host-side numpy like the reference's dataset workers)
    def generate_label(self, **kwargs) -> dict:
        """Anchor targets of one sample: ``gt_box_center`` (max_num, 7) [x, y, z, h, w, l, yaw], ``anchors`` (H, W, A, 7),
        ``mask`` (max_num) -> ``pos_equal_one`` / ``neg_equal_one`` (H, W, A), ``targets`` (H, W, 7 A).

        An anchor is positive for a box when the IoU of their axis-aligned ("stand-up") footprints exceeds ``pos_threshold`` or
        when it is the best anchor of that box (IoU > 0); negative when its IoU with every box is below ``neg_threshold``
        (best anchors excepted).  IoU with the legacy ``+ 1`` on widths and heights (``utils/box_overlaps.pyx:17-57``).  Where
        several boxes claim an anchor the reference's ``np.unique(..., return_index=True)`` keeps the first claim in the order
        [threshold matches in row-major (anchor, box) order, then best-anchor matches in box order]; reproduced with a stable sort."""
        if self.params["order"] != "hwl":
            raise AssertionError("Currently Voxel only supporthwl bbx order.")
        gt_all = np.asarray(kwargs["gt_box_center"])
        anchors = np.asarray(kwargs["anchors"])
        valid = np.asarray(kwargs["mask"]) == 1
        fmap = anchors.shape[:2]
        A = self.anchor_num
        an = anchors.reshape(-1, 7)
        diag = np.sqrt(an[:, 4] ** 2 + an[:, 5] ** 2)
        pos = np.zeros((*fmap, A))
        neg = np.zeros((*fmap, A))
        targets = np.zeros((*fmap, A * 7))

        def standup(boxes):
            c = boxes_to_corners_3d(boxes)[:, :4, :2]
            return np.concatenate([c.min(1), c.max(1)], 1).astype(np.float32)

        sa, sg = standup(an), standup(gt_all[valid])
        iw = np.minimum(sa[:, None, 2], sg[None, :, 2]) - np.maximum(sa[:, None, 0], sg[None, :, 0]) + np.float32(1)
        ih = np.minimum(sa[:, None, 3], sg[None, :, 3]) - np.maximum(sa[:, None, 1], sg[None, :, 1]) + np.float32(1)
        area_a = (sa[:, 2] - sa[:, 0] + np.float32(1)) * (sa[:, 3] - sa[:, 1] + np.float32(1))
        area_g = (sg[:, 2] - sg[:, 0] + np.float32(1)) * (sg[:, 3] - sg[:, 1] + np.float32(1))
        inter = iw * ih
        iou = np.where((iw > 0) & (ih > 0), inter / (area_a[:, None] + area_g[None] - inter), np.float32(0)).astype(np.float32)

        n_gt = iou.shape[1]
        best = iou.argmax(0) if n_gt else np.zeros(0, np.int64)
        best_gt = np.arange(n_gt)
        has = iou[best, best_gt] > 0 if n_gt else np.zeros(0, bool)
        best, best_gt = best[has], best_gt[has]
        thr_a, thr_g = np.where(iou > self.params["target_args"]["pos_threshold"])
        cand_a = np.concatenate([thr_a, best])
        cand_g = np.concatenate([thr_g, best_gt])
        order = np.argsort(cand_a, kind="stable")
        first = np.ones(len(order), bool)
        first[1:] = cand_a[order][1:] != cand_a[order][:-1]
        id_pos, id_pos_gt = cand_a[order][first], cand_g[order][first]
        id_neg = np.where((iou < self.params["target_args"]["neg_threshold"]).all(1))[0]

        ix, iy, iz = np.unravel_index(id_pos, (*fmap, A))
        pos[ix, iy, iz] = 1
        # NOTE: like the reference, the box is looked up in the UNFILTERED gt_box_center by its index among the valid boxes
        g, a_, d = gt_all[id_pos_gt], an[id_pos], diag[id_pos]
        targets[ix, iy, iz * 7 + 0] = (g[:, 0] - a_[:, 0]) / d
        targets[ix, iy, iz * 7 + 1] = (g[:, 1] - a_[:, 1]) / d
        targets[ix, iy, iz * 7 + 2] = (g[:, 2] - a_[:, 2]) / a_[:, 3]
        targets[ix, iy, iz * 7 + 3] = np.log(g[:, 3] / a_[:, 3])
        targets[ix, iy, iz * 7 + 4] = np.log(g[:, 4] / a_[:, 4])
        targets[ix, iy, iz * 7 + 5] = np.log(g[:, 5] / a_[:, 5])
        targets[ix, iy, iz * 7 + 6] = g[:, 6] - a_[:, 6]
        ix, iy, iz = np.unravel_index(id_neg, (*fmap, A))
        neg[ix, iy, iz] = 1
        ix, iy, iz = np.unravel_index(best, (*fmap, A))
        neg[ix, iy, iz] = 0
        return {"pos_equal_one": pos, "neg_equal_one": neg, "targets": targets}

    # voxel_postprocessor.py:196-229
    @staticmethod
    def collate_batch(label_batch_list) -> dict:
        stack = lambda k: torch.from_numpy(np.array([lab[k] for lab in label_batch_list]))
        return {"targets": stack("targets"), "pos_equal_one": stack("pos_equal_one"), "neg_equal_one": stack("neg_equal_one")}

    # voxel_postprocessor.py:232-352
    def post_process(self, data_dict: dict, output_dict: dict):
        corners_all, scores_all, index_all = [], [], []
        offset = 0
        for cav_id, cav_content in data_dict.items():
            if cav_id not in output_dict:
                continue
            psm = _dev_f32(output_dict[cav_id]["psm"])
            rm = _dev_f32(output_dict[cav_id]["rm"])
            if psm.shape[0] != 1:
                raise AssertionError("during validation/testing, the batch size should be 1")
            dev = psm.device
            anchors = torch.as_tensor(np.asarray(cav_content["anchor_box"]), dtype=torch.float32).to(dev).contiguous()
            A, H, W = psm.shape[1], psm.shape[2], psm.shape[3]
            T = None
            if "no_post_projection" not in cav_content:
                T = torch.as_tensor(np.asarray(cav_content["transformation_matrix"].cpu() if torch.is_tensor(
                    cav_content["transformation_matrix"]) else cav_content["transformation_matrix"]),
                    dtype=torch.float32).to(dev).contiguous()
            cap = H * W * A
            corners = torch.empty(cap, 8, 3, device=dev)
            scores = torch.empty(cap, device=dev)
            index = torch.empty(cap, device=dev, dtype=torch.int32)
            count = torch.zeros(1, device=dev, dtype=torch.int32)
            _lib.check(_lib.lib.hmvit_box_decode(
                psm.data_ptr(), rm.data_ptr(), anchors.data_ptr(), T.data_ptr() if T is not None else None, H, W, A,
                float(self.params["target_args"]["score_threshold"]), 1 if self.params["order"] == "hwl" else 0,
                corners.data_ptr(), scores.data_ptr(), index.data_ptr(), count.data_ptr(), cap, _stream()), "box_decode")
            n = int(count.item())
            if n:
                corners_all.append(corners[:n])
                scores_all.append(scores[:n])
                index_all.append(index[:n] + offset)   # ties between agents: earlier agent first, as in the vstack
            offset += cap
        if not corners_all:
            return None, None
        corners = torch.cat(corners_all).contiguous()
        scores = torch.cat(scores_all).contiguous()
        index = torch.cat(index_all).contiguous()
        n = corners.shape[0]
        dev = corners.device
        ws_bytes = int(_lib.lib.hmvit_nms_workspace_bytes(n))
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        keep = torch.empty(min(n, 1000), device=dev, dtype=torch.int32)
        n_keep = torch.zeros(1, device=dev, dtype=torch.int32)
        rng = (ctypes.c_float * 4)(GT_RANGE[0], GT_RANGE[1], GT_RANGE[3], GT_RANGE[4])
        _lib.check(_lib.lib.hmvit_nms_rotated(corners.data_ptr(), scores.data_ptr(), index.data_ptr(), n,
                                              float(self.params["nms_thresh"]), rng, ws.data_ptr(), ws_bytes,
                                              keep.data_ptr(), n_keep.data_ptr(), _stream()), "nms_rotated")
        k = keep[: int(n_keep.item())].long()
        return corners[k], scores[k]

    # voxel_postprocessor.py:355-396, for callers that want the raw decoded boxes
    @staticmethod
    def delta_to_boxes3d(deltas: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
        N = deltas.shape[0]
        deltas = deltas.permute(0, 2, 3, 1).contiguous().view(N, -1, 7)
        an = anchors.view(-1, 7).float().to(deltas.device)
        d = torch.sqrt(an[:, 4] ** 2 + an[:, 5] ** 2)
        out = torch.zeros_like(deltas)
        out[..., 0] = deltas[..., 0] * d + an[:, 0]
        out[..., 1] = deltas[..., 1] * d + an[:, 1]
        out[..., 2] = deltas[..., 2] * an[:, 3] + an[:, 2]
        out[..., 3:6] = torch.exp(deltas[..., 3:6]) * an[:, 3:6]
        out[..., 6] = deltas[..., 6] + an[:, 6]
        return out


# ---- opencood/utils/eval_utils.py ----

def voc_ap(rec, prec):
    """VOC 2010 average precision (eval_utils.py:11-34)."""
    rec, prec = list(rec), list(prec)
    rec.insert(0, 0.0); rec.append(1.0)
    mrec = rec[:]
    prec.insert(0, 0.0); prec.append(0.0)
    mpre = prec[:]
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    ap = 0.0
    for i in range(1, len(mrec)):
        if mrec[i] != mrec[i - 1]:
            ap += (mrec[i] - mrec[i - 1]) * mpre[i]
    return ap, mrec, mpre


def caluclate_tp_fp(det_boxes, det_score, gt_boxes, result_stat, iou_thresh):
    """eval_utils.py:144-196 (mode 'iou'): detections in descending score order, each takes the remaining GT box of
    largest IoU if that IoU reaches the threshold."""
    fp, tp = [], []
    gt = gt_boxes.shape[0]
    if det_boxes is not None:
        iou = quad_iou(det_boxes, gt_boxes).cpu().numpy() if gt else np.zeros((det_boxes.shape[0], 0), np.float32)
        order = np.argsort(-det_score.detach().cpu().numpy())
        remaining = list(range(gt))
        for i in order:
            ious = iou[i, remaining]
            if len(remaining) == 0 or np.max(ious) < iou_thresh:
                fp.append(1); tp.append(0)
                continue
            fp.append(0); tp.append(1)
            remaining.pop(int(np.argmax(ious)))
    result_stat[iou_thresh]["fp"] += fp
    result_stat[iou_thresh]["tp"] += tp
    result_stat[iou_thresh]["gt"] += gt


def calculate_ap(result_stat, iou):
    """eval_utils.py:199-237 (cumulative sums in the order the frames were appended, as the reference does)."""
    st = result_stat[iou]
    fp, tp = list(st["fp"]), list(st["tp"])
    assert len(fp) == len(tp)
    gt_total = st["gt"]
    fp, tp = np.cumsum(fp).tolist(), np.cumsum(tp).tolist()
    rec = [float(t) / gt_total for t in tp]
    prec = [float(t) / (f + t) for t, f in zip(tp, fp)]
    return voc_ap(rec[:], prec[:])
