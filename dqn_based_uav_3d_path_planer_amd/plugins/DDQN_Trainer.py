"""Drop-in for Trainer/DDQN_Trainer.py: argmax from q_local, value from q_target (DDQN_Trainer.py:93-99)."""
from _trainer_base import BaseDQNTrainer


class DDQN_Trainer(BaseDQNTrainer):
    KIND = "ddqn"
    FILE_TAG = "DDQN_"
