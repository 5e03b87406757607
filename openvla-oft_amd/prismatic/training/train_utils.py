"""Action-token masks and token metrics (mirror of prismatic/training/train_utils.py:8-56).  Integer host logic on the
label tensors; used by the glue for logging -- the model builds the same masks on device inside ovla_assemble_multimodal."""
import torch

from ..vla import constants as C


def get_current_action_mask(token_ids):
    cumsum = torch.cumsum(token_ids != C.IGNORE_INDEX, dim=1)
    return (token_ids > C.ACTION_TOKEN_BEGIN_IDX) * ((1 <= cumsum) & (cumsum <= C.ACTION_DIM))


def get_next_actions_mask(token_ids):
    cumsum = torch.cumsum(token_ids != C.IGNORE_INDEX, dim=1)
    return (token_ids > C.ACTION_TOKEN_BEGIN_IDX) * (cumsum > C.ACTION_DIM)


def compute_token_accuracy(predicted_token_ids, ground_truth_token_ids, mask):
    correct = (predicted_token_ids == ground_truth_token_ids) & mask
    return correct.sum().float() / mask.sum().float()


def compute_actions_l1_loss(action_tokenizer, predicted_token_ids, ground_truth_token_ids, mask):
    pred = torch.tensor(action_tokenizer.decode_token_ids_to_actions(predicted_token_ids[mask].cpu().numpy()))
    true = torch.tensor(action_tokenizer.decode_token_ids_to_actions(ground_truth_token_ids[mask].cpu().numpy()))
    return torch.nn.functional.l1_loss(pred, true)
