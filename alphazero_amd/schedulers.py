"""Temperature schedulers (schedulers.py:4-46)."""
from .base import TemperatureScheduler


class ConstantTemperatureScheduler(TemperatureScheduler):
    def __init__(self, temp_max_step, temp_min_step, max_steps):
        super().__init__(temp_max_step, temp_min_step, max_steps)
        if self.temp_min_step != self.temp_max_step:
            raise ValueError("temp_min_step should be equal to temp_max_step for constant scheduler.")

    def compute_temperature(self, step):
        return 0  # the reference's constant scheduler is always greedy (schedulers.py:15-17)


class LinearTemperatureScheduler(TemperatureScheduler):
    """1 up to temp_max_step, 0 from temp_min_step on, linear in between"""

    def __init__(self, temp_max_step, temp_min_step, max_steps):
        super().__init__(temp_max_step, temp_min_step, max_steps)
        if self.temp_min_step < self.temp_max_step:
            raise ValueError("temp_min_step should be greater than temp_max_step for linear scheduler.")

    def compute_temperature(self, step):
        if step <= self.temp_max_step:
            return 1
        if step >= self.temp_min_step:
            return 0
        return 1 - (step - self.temp_max_step) / (self.temp_min_step - self.temp_max_step)


TEMP_SCHEDULERS = {"constant": ConstantTemperatureScheduler, "linear": LinearTemperatureScheduler}
