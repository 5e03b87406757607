"""Action-token masks and token metrics with the reference's names and semantics (prismatic/training/train_utils.py:8-56).

Integer host logic on the label tensors, used by the glue for logging; the model builds the same masks on the device inside
ovla_assemble_multimodal.  A position belongs to the action chunk when its label is an action token (id above
ACTION_TOKEN_BEGIN_IDX); its ordinal among the non-ignored labels of the row (1-based) tells whether it is part of the current
action (ordinals 1 .. ACTION_DIM) or of the following ones."""
import torch

from ..vla import constants as C


def _ordinal_and_action(token_ids: torch.Tensor):
    counted = token_ids.ne(C.IGNORE_INDEX)
    return counted.cumsum(dim=1), token_ids.gt(C.ACTION_TOKEN_BEGIN_IDX)


def get_current_action_mask(token_ids: torch.Tensor) -> torch.Tensor:
    ordinal, is_action = _ordinal_and_action(token_ids)
    return is_action & ordinal.ge(1) & ordinal.le(C.ACTION_DIM)


def get_next_actions_mask(token_ids: torch.Tensor) -> torch.Tensor:
    ordinal, is_action = _ordinal_and_action(token_ids)
    return is_action & ordinal.gt(C.ACTION_DIM)


def compute_token_accuracy(predicted_token_ids, ground_truth_token_ids, mask):
    hits = predicted_token_ids.eq(ground_truth_token_ids) & mask
    return hits.sum().float() / mask.sum().float()


def compute_actions_l1_loss(action_tokenizer, predicted_token_ids, ground_truth_token_ids, mask):
    decode = action_tokenizer.decode_token_ids_to_actions
    predicted = torch.tensor(decode(predicted_token_ids[mask].cpu().numpy()))
    target = torch.tensor(decode(ground_truth_token_ids[mask].cpu().numpy()))
    return torch.nn.functional.l1_loss(predicted, target)
