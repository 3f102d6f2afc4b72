"""Drop-in for Trainer/DuelingDQN_Trainer.py: VAnet2 dueling head + double-DQN target."""
from _trainer_base import BaseDQNTrainer


class DuelingDQN_Trainer(BaseDQNTrainer):
    KIND = "dueling"
    FILE_TAG = "DuelingDQN_"
