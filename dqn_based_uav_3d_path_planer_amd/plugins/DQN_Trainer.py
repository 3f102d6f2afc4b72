"""Drop-in for Trainer/DQN_Trainer.py: max_a Q_target(s') target, MSE, Adam, hard copy every Update_loop."""
from _trainer_base import BaseDQNTrainer


class DQN_Trainer(BaseDQNTrainer):
    KIND = "dqn"
    FILE_TAG = ""
