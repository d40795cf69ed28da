"""Epoch loop around VanGan.train_step: host mirror of the reference's `train()` (vangan.py:510-551), the epoch loop of
main.py:214-236 and the scalar part of GanMonitor (custom_callback.py:326-424, :426-445).  SURVEY section 8 row (f)3.

Nothing here touches the device: the monitor writes two host scalars (`VanGan.lr`, `VanGan.layer_noise`) that the step
reads when it enqueues Adam and the discriminator noise, the loop pulls batches from `data.DataPipeline` (which keeps the
volumes and produces the batches in HBM)."""
from __future__ import annotations

from typing import Dict, List, Optional

from . import data


class GanMonitor:
    """Schedules only (plots / TensorBoard of the reference's GanMonitor are out of scope)."""

    def __init__(self, EPOCHS: int, INITIATE_LR_DECAY: int, INITIAL_LR: float, train_steps: int, NO_NOISE: int,
                 init_noise: Optional[float] = None, schedule_step: str = 'global_iterations'):
        """schedule_step: 'global_iterations' (default) evaluates the decay at the optimizer's iteration count as Keras
        does -- reference-exact, and with the reference defaults the rate is 0 from epoch INITIATE_LR_DECAY on;
        'since_install' is the linear decay the reference presumably intended (data.learning_rate)."""
        if schedule_step not in ('global_iterations', 'since_install'):
            raise ValueError("schedule_step must be 'global_iterations' or 'since_install'")
        self.EPOCHS, self.INITIATE_LR_DECAY, self.INITIAL_LR = EPOCHS, INITIATE_LR_DECAY, INITIAL_LR
        self.train_steps, self.NO_NOISE = train_steps, NO_NOISE
        self.init_noise = init_noise
        self.schedule_step = schedule_step
        self.resume_epoch: Optional[int] = None

    @staticmethod
    def _iterations(model) -> Optional[int]:
        """optimizer.iterations of the reference = Adam steps applied so far (all four optimizers step together)."""
        stores = getattr(model, 'stores', None)
        return None if not stores else int(next(iter(stores.values())).step)

    def set_learning_rate(self, model, epoch: int, step_in_epoch: int = 0):
        """custom_callback.py:326-397.  The reference installs a PolynomialDecay object once and lets the optimizer's
        iteration counter drive it; here the same value is written before every step."""
        if getattr(model, 'checkpoint_loaded', False) and epoch > self.INITIATE_LR_DECAY:
            model.checkpoint_loaded = False
            self.resume_epoch = epoch
        it = self._iterations(model)
        if self.resume_epoch is not None:
            model.lr = data.learning_rate_resumed(self.INITIAL_LR, self.resume_epoch, epoch, step_in_epoch, self.EPOCHS,
                                                  self.INITIATE_LR_DECAY, self.train_steps, self.schedule_step, it)
        else:
            model.lr = data.learning_rate(self.INITIAL_LR, epoch, step_in_epoch, self.EPOCHS, self.INITIATE_LR_DECAY,
                                          self.train_steps, self.schedule_step, it)
        return model.lr

    def updateDiscriminatorNoise(self, model, epoch: int):
        if self.init_noise is None:
            self.init_noise = model.layer_noise                    # custom_callback.py:441 passes model.layer_noise
        model.layer_noise = data.discriminator_noise(self.init_noise, epoch, self.NO_NOISE)
        return model.layer_noise

    def on_epoch_start(self, model, epoch: int):
        """custom_callback.py:426-445."""
        self.set_learning_rate(model, epoch, 0)
        self.updateDiscriminatorNoise(model, epoch)


class NonFiniteLoss(FloatingPointError):
    """A result scalar of a step is NaN / inf.  The reference has no such check (its check_numerics calls are commented out,
    vangan.py:290-292) and min_max_norm_tf has no epsilon (utils.py:48): one constant patch turns every weight into NaN silently."""


def train(ds, gan, epoch: int, steps: Optional[int] = None, training: bool = True,
          monitor: Optional[GanMonitor] = None, on_nonfinite: str = 'raise') -> Dict[str, List[float]]:
    """vangan.py:510-551: `steps` batches through distributed_train_step / distributed_test_step, results appended per
    key.  `ds` is an iterable of (real_I, real_S) or an object with next_batch().
    on_nonfinite: 'raise' (default; NonFiniteLoss names the step and the keys -- SURVEY section 5's failure-detection hook) or
    'ignore' (the reference's behaviour: NaNs are appended and training goes on)."""
    if on_nonfinite not in ('raise', 'ignore'):
        raise ValueError("on_nonfinite must be 'raise' or 'ignore'")
    results: Dict[str, List[float]] = {}
    it = iter(ds) if hasattr(ds, '__iter__') else None
    cntr = 0
    while steps is None or cntr < steps:
        if it is not None:
            try:
                x, y = next(it)
            except StopIteration:
                break
        else:
            x, y = ds.next_batch()
        if training:
            if monitor is not None:
                monitor.set_learning_rate(gan, epoch, cntr)
            result = gan.distributed_train_step(x, y)
        else:
            result = gan.distributed_test_step(x, y)
        cntr += 1
        for k, v in result.items():
            results.setdefault(k, []).append(v)
        if on_nonfinite == 'raise':
            bad = [k for k, v in result.items() if v != v or v in (float('inf'), float('-inf'))]
            if bad:
                raise NonFiniteLoss('epoch %d, step %d (%s): non-finite %s -- a constant patch makes min_max_norm divide by zero '
                                    '(utils.py:48); the weights of this step are already updated, resume from the last checkpoint'
                                    % (epoch, cntr - 1, 'train' if training else 'validation', ', '.join(bad)))
    return results


def fit(gan, train_ds, monitor: GanMonitor, val_ds=None, val_steps: int = 0, start_epoch: int = 0,
        checkpoint_period: int = 2, save: bool = True):
    """main.py:214-236 without the plots: per epoch the monitor's schedules, train_steps training steps, val_steps
    validation steps, a checkpoint when epoch % PERIOD == 1 or at the last epoch."""
    history = []
    for epoch in range(start_epoch, monitor.EPOCHS):
        gan.current_epoch = epoch
        monitor.on_epoch_start(gan, epoch)
        tr = train(train_ds, gan, epoch, monitor.train_steps, True, monitor)
        va = train(val_ds, gan, epoch, val_steps, False) if val_ds is not None and val_steps else {}
        if save and (epoch % checkpoint_period == 1 or epoch == monitor.EPOCHS - 1):
            gan.save_checkpoint(epoch=epoch)
        history.append({'epoch': epoch, 'lr': gan.lr, 'noise': gan.layer_noise,
                        'train': {k: sum(v) / len(v) for k, v in tr.items()},
                        'val': {k: sum(v) / len(v) for k, v in va.items()}})
    return history
