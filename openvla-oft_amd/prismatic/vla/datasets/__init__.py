"""Mirror of prismatic/vla/datasets/__init__.py: the data path the fine-tune script consumes, without TensorFlow (rlds_free.py)."""
from .rlds_free import (DEFAULT_AUGMENT_KWARGS, DeviceCollator, Episode, EpisodeDataset, RLDSBatchTransform, batches, build_prompt,
                        chunk_indices, get_dataset_statistics, list_episodes, normalize_action_and_proprio, sample_augment_params,
                        write_episode)

RLDSDataset = EpisodeDataset   # the name finetune.py imports (prismatic/vla/datasets/datasets.py:100)
