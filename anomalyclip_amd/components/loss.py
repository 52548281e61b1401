"""ComputeLoss -- drop-in for the reference's `src.models.components.loss.ComputeLoss` (loss.py:20-195):
same constructor keys (configs/model/*.yaml `loss:` block), same call signature, same 8-tuple result.
All seven terms and their gradients are evaluated by ONE libacx kernel pass (acx_mil_loss) instead of
the reference's Python loop over classes with `.nonzero()` host syncs."""
from __future__ import annotations

from .functional import MilLossFn


class ComputeLoss:
    def __init__(self, normal_id, num_topk, lambda_dir_abn, lambda_dir_nor, lambda_topk_abn, lambda_bottomk_abn,
                 lambda_topk_nor, lambda_smooth, lambda_sparse, frames_per_segment, num_segments):
        self.normal_id, self.num_topk = normal_id, num_topk
        self.lambda_dir_abn, self.lambda_dir_nor = lambda_dir_abn, lambda_dir_nor
        self.lambda_topk_abn, self.lambda_bottomk_abn, self.lambda_topk_nor = lambda_topk_abn, lambda_bottomk_abn, lambda_topk_nor
        self.lambda_smooth, self.lambda_sparse = lambda_smooth, lambda_sparse
        self.frames_per_segment, self.num_segments = frames_per_segment, num_segments

    def __call__(self, similarity, similarity_topk, labels, scores, idx_topk_abn, idx_topk_nor, idx_bottomk_abn):
        lambdas = (self.lambda_dir_abn, self.lambda_dir_nor, self.lambda_topk_abn, self.lambda_bottomk_abn,
                   self.lambda_topk_nor, self.lambda_smooth, self.lambda_sparse)
        cfg = (self.num_segments, self.frames_per_segment, self.num_topk, self.normal_id, lambdas)
        return MilLossFn.apply(similarity, similarity_topk, scores, labels, idx_topk_abn, idx_topk_nor, idx_bottomk_abn, cfg)
