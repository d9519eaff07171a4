"""CPU: the epoch/step bookkeeping of a training run (grappa_amd/schedule.py) against the behaviour the reference's LitModel
specifies (training/lightning_model.py:60, :123-151, :181-201, :246-293; pytorch_lightning is absent offline, so the
expectations are worked out from those lines by hand)."""
import types

from grappa_amd.schedule import TrainSchedule


def test_warmup_restarts_and_loss_weights():
    s = TrainSchedule(lr=1e-3, start_qm_epochs=2, add_restarts=[5], warmup_steps=4, energy_weight=1., gradient_weight=0.8, param_weight=1e-3,
                      param_loss_epochs=7)
    assert s.restarts == [2, 5, 7]
    assert s.initial_loss_weights() == dict(gradient_weight=0, energy_weight=0, param_weight=1e-3, tuplewise_weight=0.)
    loss = types.SimpleNamespace(**s.initial_loss_weights())
    resets = []
    opt = types.SimpleNamespace(lr=None, reset_state=lambda: resets.append(1))
    # epochs 0, 1: no restart, constant lr, parameter loss only
    for ep in (0, 1):
        assert s.on_train_epoch_start(ep, loss, opt) is False
        assert [s.next_lr() for _ in range(3)] == [1e-3] * 3
        assert (loss.energy_weight, loss.gradient_weight, loss.param_weight) == (0, 0, 1e-3)
    # epoch 2 = start of the QM epochs: optimiser restart, linear warm-up from 0 over 4 steps, then lr; configured weights
    assert s.on_train_epoch_start(2, loss, opt) is True and resets == [1] and opt.lr == 1e-3
    assert [s.next_lr() for _ in range(6)] == [0.0, 0.25e-3, 0.5e-3, 0.75e-3, 1e-3, 1e-3]
    assert (loss.energy_weight, loss.gradient_weight, loss.param_weight) == (1., 0.8, 1e-3)
    assert s.on_train_epoch_start(3, loss, opt) is False and s.next_lr() == 1e-3
    # epoch 5: additional restart
    assert s.on_train_epoch_start(5, loss, opt) is True and s.next_lr() == 0.0
    # epoch 7: parameter loss switched off, optimiser restarted again
    assert s.on_train_epoch_start(7, loss, opt) is True and loss.param_weight == 0. and len(resets) == 3


def test_lr_decay_on_early_stopping_metric():
    s = TrainSchedule(lr=1.0, start_qm_epochs=0, patience=2, lr_decay=0.5, early_stopping_energy_weight=3.)
    m = lambda e, f: {"avg": {"rmse_energies": e, "rmse_gradients": f}}   # noqa: E731
    assert s.on_validation_epoch_end(0, m(1., 1.)) is None            # only epochs > start_qm_epochs count (:257)
    assert s.on_validation_epoch_end(1, m(1., 1.)) == 4.0 and s.best_early_stopping_loss == 4.0
    for ep in (2, 3):                                                 # two epochs without improvement: still within patience
        s.on_validation_epoch_end(ep, m(1., 2.))
    assert s.lr == 1.0 and s.epochs_without_improvement == 2
    s.on_validation_epoch_end(4, m(1., 2.))                           # third: lr decays, the best value is RESET to the current one
    assert s.lr == 0.5 and s.epochs_without_improvement == 0 and s.best_early_stopping_loss == 5.0
    s.on_validation_epoch_end(5, m(1., 1.5))
    assert s.best_early_stopping_loss == 4.5 and s.lr == 0.5
    # finish criterion: stop if the metric exceeds the bound that applies after the elapsed time
    s2 = TrainSchedule(start_qm_epochs=0, finish_criterion={-1.0: 3.0})
    s2.on_validation_epoch_end(1, m(1., 2.))
    assert s2.should_stop
