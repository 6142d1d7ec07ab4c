"""Learning drivers (loop owner, recording, abort guards) -- mirror of ``tc_gan/drivers.py`` for the
BPTT (c)WGAN path.  No arithmetic: they consume the `info` namespaces of ``gan.learning()``."""
from logging import getLogger
import collections

import numpy as np

from . import execution, param_file, ssnode
from .recorders import (ConditionalTuningCurveStatsRecorder, DiscLearningRecorder, DiscParamStatsRecorder,
                        FlexGenParamRecorder, GenMomentsRecorder, LearningRecorder, MMLearningRecorder, _host)
from .utils import Namespace

logger = getLogger(__name__)


def is_at_interval(step, interval):
    return interval > 0 and step % interval == 0


def net_isfinite(discriminator):
    return all(np.isfinite(arr).all() for arr in discriminator.get_param_values())


class recording_exit_reason(object):
    """``with recording_exit_reason(datastore): <loop>`` -- what ended the loop goes to exit.json
    (drivers.py:31-58): a clean end is 'end_of_iteration' (good), Ctrl-C 'keyboard_interrupt', any other exception
    'uncaught_exception' with its text.  A KnownError passes through untouched: whoever raised it has already written
    its own exit.json.  Exceptions are never swallowed."""

    def __init__(self, datastore):
        self.datastore = datastore

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.datastore.save_exit_reason(reason='end_of_iteration', good=True)
        elif issubclass(exc_type, execution.KnownError):
            pass
        elif issubclass(exc_type, KeyboardInterrupt):
            self.datastore.save_exit_reason(reason='keyboard_interrupt', good=False)
        elif issubclass(exc_type, Exception):
            self.datastore.save_exit_reason(reason='uncaught_exception', good=False, exception=str(exc))
        return False


def _abort(datastore, reason, message, exit_code, **details):
    """An expected abort: exit.json names the reason (good=False), the process ends with `exit_code`."""
    datastore.dump_json(dict(reason=reason, good=False, **details), 'exit.json')
    raise execution.KnownError(message, exit_code=exit_code)


def maybe_quit(datastore, JDS_fake, JDS_true, quit_JDS_threshold):
    """Give up when the generator has wandered too far from the true parameters (drivers.py:183-198): Euclidean
    distance over all twelve entries of (J, D, S); a threshold <= 0 switches the check off."""
    if not quit_JDS_threshold > 0:
        return
    gap = np.concatenate(JDS_fake).ravel() - np.concatenate(JDS_true).ravel()
    JDS_distance = np.linalg.norm(gap)
    if JDS_distance >= quit_JDS_threshold:
        _abort(datastore, 'JDS_distance',
               'Exit simulation since (J, D, S)-distance (= {}) to the true parameter exceed threshold (= {}).'
               .format(JDS_distance, quit_JDS_threshold), 4, JDS_distance=JDS_distance)


def check_disc_param(datastore, discriminator, nnorms):
    """A critic that has gone NaN/inf ends the run with exit code 3 (drivers.py:201-211).  The cheap test is on the
    recorded norms; only when one of them is not finite are the parameters themselves read back and checked."""
    finite = np.isfinite(nnorms)
    if finite.all() or net_isfinite(discriminator):
        return
    _abort(datastore, 'disc_param_has_nan', "Discriminator parameter is not finite.", 3,
           isfinite_nnorms=finite.tolist())


class SSNRejectionLimiter(object):
    """Abort when the fixed-point finder keeps rejecting most of its draws (drivers.py:214-255): the share
    rejections / (rejections + n_samples) must stay at or below `rejection_limit`; more than
    `max_consecutive_exceedings` violations IN A ROW end the run (one good step resets the count)."""

    def __init__(self, datastore, n_samples, rejection_limit=0.6, max_consecutive_exceedings=5):
        self.datastore = datastore
        self.n_samples = n_samples
        self.rejection_limit = rejection_limit
        self.max_consecutive_exceedings = max_consecutive_exceedings
        self._exceedings = 0

    def should_abort(self, rejections):
        share = rejections / (rejections + self.n_samples)
        self._exceedings = self._exceedings + 1 if share > self.rejection_limit else 0
        return self._exceedings > self.max_consecutive_exceedings

    def __call__(self, rejections):
        if self.should_abort(rejections):
            _abort(self.datastore, 'too_many_rejections', "Too many rejections in fixed-point finder.", 4)

    @classmethod
    def from_driver(cls, driver):
        return cls(driver.datastore, n_samples=driver.gan.NZ)


class WGANDiscLossLimiter(object):
    """Abort when the critic loss has blown up (drivers.py:265-295): over a sliding window of the last `hist_length`
    critic losses -- judged only once the window is full -- more than `prob_limit` of them exceed `wild_disc_loss`.
    As in the reference the comparison is one-sided (its abs() wraps the comparison's result, not the loss): hugely
    NEGATIVE losses do not count."""

    def __init__(self, datastore, prob_limit=0.6, wild_disc_loss=10000, hist_length=50):
        self.datastore = datastore
        self.prob_limit = prob_limit
        self.wild_disc_loss = wild_disc_loss
        self.hist_length = hist_length
        self.dloss_hist = collections.deque(maxlen=hist_length)

    def prob_exceed(self):
        window = np.asarray(self.dloss_hist)
        return np.count_nonzero(window > self.wild_disc_loss) / len(window)

    def should_abort(self, dloss):
        self.dloss_hist.append(dloss)
        return len(self.dloss_hist) == self.hist_length and self.prob_exceed() > self.prob_limit

    def __call__(self, dloss):
        if self.should_abort(dloss):
            _abort(self.datastore, 'wild_disc_loss', "Too many wild discriminator losses.", 4)

    @classmethod
    def from_driver(cls, driver):
        return cls(driver.datastore)


class _GeneratorStepLoop(object):
    """The part of a driver that does not depend on the learner: a counted loop over generator steps with the exit
    reason recorded, and the per-step bookkeeping both learners need -- log the generator's parameters, flush the
    tables, give up when the parameters have drifted beyond `quit_JDS_threshold`.

    Subclasses provide `open_recorders()` and `record_step(gen_step, result)`; the reference's method names (`pre_loop`,
    `post_update`, `iterate`) stay available because its run scripts and subclasses use them (drivers.py:97-180)."""

    start_step = 0
    quit_JDS_threshold = -1

    def watch_generator(self, gen_step):
        """Log (J, D, S[, V]) of this step and apply the distance guard.  NB: the reference exponentiates the recorded
        (J, D, S) before comparing them with ssnode's default parameters -- a leftover from when they were stored as
        logarithms (drivers.py:147-152); kept so that a run ends exactly where the reference's would."""
        recorded = self.generator_recorder.record(gen_step)
        self.datastore.flush_all()
        maybe_quit(self.datastore, JDS_fake=[np.exp(block) for block in recorded],
                   JDS_true=[ssnode.DEFAULT_PARAMS[name] for name in 'JDS'], quit_JDS_threshold=self.quit_JDS_threshold)

    def iterate(self, update_func):
        """`update_func(gen_step)` advances the learner by one generator step and returns what `post_update` records."""
        self.pre_loop()
        name = type(self).__name__
        logger.info('%s: start iterations', name)
        with recording_exit_reason(self.datastore):
            for gen_step in range(self.start_step, self.iterations):
                self.post_update(gen_step, update_func(gen_step))
        logger.info('%s: maximum iterations reached', name)


class BPTTWGANDriver(_GeneratorStepLoop):
    """Driver of the BPTT Wasserstein GANs (drivers.py:61-180 + 298-337): consumes `gan.learning()`, one critic or
    generator update per item, and records / guards after each."""

    def __init__(self, gan, datastore, iterations, quiet, disc_param_save_interval, disc_param_template,
                 disc_param_save_on_error, quit_JDS_threshold=-1, checkpoint_interval=-1, resume_from=None, **kwargs):
        self.gan, self.datastore = gan, datastore
        self.iterations, self.quiet = iterations, quiet
        self.disc_param_save_interval = disc_param_save_interval
        self.disc_param_template = disc_param_template
        self.disc_param_save_on_error = disc_param_save_on_error
        self.quit_JDS_threshold = quit_JDS_threshold
        self.checkpoint_interval = checkpoint_interval      # new: write <datastore>/checkpoint.pkl every K generator steps
        self.resume_from = resume_from                      # new: continue from such a file
        self.__dict__.update(kwargs)

    def _disc_param_path(self, name):
        return self.datastore.path('disc_param', name)

    def pre_loop(self):
        for attr, recorder in (('learning_recorder', LearningRecorder), ('generator_recorder', FlexGenParamRecorder),
                               ('discparamstats_recorder', DiscParamStatsRecorder),
                               ('disclearning_recorder', DiscLearningRecorder),
                               ('rejection_limiter', SSNRejectionLimiter), ('disc_loss_limiter', WGANDiscLossLimiter)):
            setattr(self, attr, recorder.from_driver(self))

    def post_disc_update(self, gen_step, disc_step, Dloss, Daccuracy, SSsolve_time, gradient_time, model_info):
        """After every critic update: its row, the parameter statistics, then the three guards (critic went non-finite,
        finder rejects too much, loss blew up)."""
        self.disclearning_recorder.record(gen_step, disc_step, Dloss, Daccuracy, SSsolve_time, gradient_time,
                                          model_info.rejections, model_info.unused)
        check_disc_param(self.datastore, self.gan.discriminator, self.discparamstats_recorder.record(gen_step, disc_step))
        self.rejection_limiter(model_info.rejections)
        self.disc_loss_limiter(Dloss)

    def post_update(self, gen_step, update_result):
        """After every generator update: its row, periodic critic snapshot / checkpoint, then `watch_generator`."""
        self.learning_recorder.record(gen_step, update_result)
        if is_at_interval(gen_step, self.disc_param_save_interval):
            param_file.dump(self.gan.discriminator, self._disc_param_path(self.disc_param_template.format(gen_step)))
        if is_at_interval(gen_step, self.checkpoint_interval):
            self.gan.save_checkpoint(self.datastore.path('checkpoint.pkl'), gen_step)
        self.watch_generator(gen_step)

    def _guarded(self, update_func):
        """`--disc-param-save-on-error`: the critic as it was before the step, and as the failing step left it."""
        def step(gen_step):
            param_file.dump(self.gan.discriminator, self._disc_param_path('pre_error.npz'))
            try:
                return update_func(gen_step)
            except Exception:
                param_file.dump(self.gan.discriminator, self._disc_param_path('post_error.npz'))
                raise
        return step

    def iterate(self, update_func):
        super(BPTTWGANDriver, self).iterate(self._guarded(update_func) if self.disc_param_save_on_error else update_func)

    def run(self, gan):
        if self.resume_from:
            self.start_step = gan.load_checkpoint(self.resume_from)
            logger.info('resumed from %s: continuing with generator step %d', self.resume_from, self.start_step)
        updates = gan.learning(self.start_step)
        last_critic = [None]

        def until_generator_update(gen_step):
            for info in updates:
                if info.is_discriminator:
                    self.post_disc_update(info.gen_step, info.disc_step, info.disc_loss, info.accuracy, info.gen_time,
                                          info.disc_time, ssnode.null_FixedPointsInfo)
                    last_critic[0] = info
                    continue
                assert info.gen_step == gen_step
                # TC_mean.csv: batch means of the generated, then of the data curves of the last critic step
                critic = last_critic[0]
                self.datastore.tables.saverow('TC_mean.csv', _host(critic.xg).mean(axis=0).tolist()
                                              + _host(critic.xd).mean(axis=0).tolist())
                return Namespace(info=info, disc_info=critic)
            raise RuntimeError('gan.learning() ended before generator step {}'.format(gen_step))
        self.iterate(until_generator_update)


class BPTTcWGANDriver(BPTTWGANDriver):
    """The conditional GAN adds tuning-curve statistics every `tc_stats_record_interval` steps (drivers.py:340-351)."""

    def pre_loop(self):
        super(BPTTcWGANDriver, self).pre_loop()
        self.tuning_curve_recorder = ConditionalTuningCurveStatsRecorder.from_driver(self)

    def post_update(self, gen_step, update_result):
        if is_at_interval(gen_step, self.tc_stats_record_interval):
            self.tuning_curve_recorder.record(gen_step, update_result.disc_info)
        super(BPTTcWGANDriver, self).post_update(gen_step, update_result)


class MomentMatchingDriver(_GeneratorStepLoop):
    """Driver of `BPTTMomentMatcher` (drivers.py:354-421): one generator update per item of `learner.learning()`."""

    def __init__(self, mmatcher, datastore, iterations, quiet, gen_moments_record_interval, quit_JDS_threshold=-1):
        self.mmatcher, self.datastore = mmatcher, datastore
        self.iterations, self.quiet = iterations, quiet
        self.gen_moments_record_interval = gen_moments_record_interval
        self.quit_JDS_threshold = quit_JDS_threshold

    gan = property(lambda self: self.mmatcher)        # FlexGenParamRecorder.from_driver looks for `.gan`

    def pre_loop(self):
        self.learning_recorder = MMLearningRecorder.from_driver(self)
        self.gen_moments_recorder = GenMomentsRecorder.from_driver(self)
        self.generator_recorder = FlexGenParamRecorder.from_driver(self)

    def post_update(self, gen_step, update_result):
        self.learning_recorder.record(gen_step, update_result)
        if is_at_interval(gen_step, self.gen_moments_record_interval):
            self.gen_moments_recorder.record(gen_step, update_result)
        self.watch_generator(gen_step)

    def run(self, learner):
        updates = learner.learning()

        def one_update(gen_step):
            info = next(updates)
            assert info.step == gen_step
            return info
        self.iterate(one_update)
