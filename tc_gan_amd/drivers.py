"""Learning drivers (loop owner, recording, abort guards) -- mirror of ``tc_gan/drivers.py`` for the
BPTT (c)WGAN path.  No arithmetic: they consume the `info` namespaces of ``gan.learning()``."""
from logging import getLogger
import collections
import contextlib

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


@contextlib.contextmanager
def recording_exit_reason(datastore):
    """drivers.py:31-58."""
    try:
        yield
    except KeyboardInterrupt:
        datastore.save_exit_reason(reason='keyboard_interrupt', good=False)
        raise
    except execution.KnownError:
        raise
    except Exception as err:
        datastore.save_exit_reason(reason='uncaught_exception', good=False, exception=str(err))
        raise
    else:
        datastore.save_exit_reason(reason='end_of_iteration', good=True)


def maybe_quit(datastore, JDS_fake, JDS_true, quit_JDS_threshold):
    """drivers.py:183-198."""
    JDS_fake = np.concatenate(JDS_fake).flatten()
    JDS_true = np.concatenate(JDS_true).flatten()
    JDS_distance = np.linalg.norm(JDS_fake - JDS_true)
    if quit_JDS_threshold > 0 and JDS_distance >= quit_JDS_threshold:
        datastore.dump_json(dict(reason='JDS_distance', JDS_distance=JDS_distance, good=False), 'exit.json')
        raise execution.KnownError(
            'Exit simulation since (J, D, S)-distance (= {}) to the true parameter exceed threshold (= {}).'
            .format(JDS_distance, quit_JDS_threshold), exit_code=4)


def check_disc_param(datastore, discriminator, nnorms):
    """drivers.py:201-211: NaN critic -> exit.json + exit code 3."""
    isfinite_nnorms = np.isfinite(nnorms)
    if not isfinite_nnorms.all() and not net_isfinite(discriminator):
        datastore.dump_json(dict(reason='disc_param_has_nan', isfinite_nnorms=isfinite_nnorms.tolist(), good=False),
                            'exit.json')
        raise execution.KnownError("Discriminator parameter is not finite.", exit_code=3)


class SSNRejectionLimiter(object):
    """drivers.py:214-255."""

    def __init__(self, datastore, n_samples, rejection_limit=0.6, max_consecutive_exceedings=5):
        self.datastore = datastore
        self.n_samples = n_samples
        self.rejection_limit = rejection_limit
        self.max_consecutive_exceedings = max_consecutive_exceedings
        self._exceedings = 0

    def should_abort(self, rejections):
        if rejections / (rejections + self.n_samples) > self.rejection_limit:
            self._exceedings += 1
        else:
            self._exceedings = 0
        return self._exceedings > self.max_consecutive_exceedings

    def __call__(self, rejections):
        if self.should_abort(rejections):
            self.datastore.dump_json(dict(reason='too_many_rejections', good=False), 'exit.json')
            raise execution.KnownError("Too many rejections in fixed-point finder.", exit_code=4)

    @classmethod
    def from_driver(cls, driver):
        return cls(driver.datastore, n_samples=driver.gan.NZ)


class WGANDiscLossLimiter(object):
    """drivers.py:265-295."""

    def __init__(self, datastore, prob_limit=0.6, wild_disc_loss=10000, hist_length=50):
        self.datastore = datastore
        self.prob_limit = prob_limit
        self.wild_disc_loss = wild_disc_loss
        self.hist_length = hist_length
        self.dloss_hist = collections.deque(maxlen=hist_length)

    def prob_exceed(self):
        return np.mean(abs(np.asarray(self.dloss_hist) > self.wild_disc_loss))

    def should_abort(self, dloss):
        self.dloss_hist.append(dloss)
        return len(self.dloss_hist) == self.hist_length and self.prob_exceed() > self.prob_limit

    def __call__(self, dloss):
        if self.should_abort(dloss):
            self.datastore.dump_json(dict(reason='wild_disc_loss', good=False), 'exit.json')
            raise execution.KnownError("Too many wild discriminator losses.", exit_code=4)

    @classmethod
    def from_driver(cls, driver):
        return cls(driver.datastore)


class BPTTWGANDriver(object):
    """drivers.py:61-180 + 298-337 (GANDriver machinery specialised to the BPTT WGANs)."""

    def __init__(self, gan, datastore, iterations, quiet, disc_param_save_interval, disc_param_template,
                 disc_param_save_on_error, quit_JDS_threshold=-1, checkpoint_interval=-1, resume_from=None, **kwargs):
        self.checkpoint_interval = checkpoint_interval      # new: write <datastore>/checkpoint.pkl every K generator steps
        self.resume_from = resume_from                      # new: continue from such a file
        self.start_step = 0
        self.gan = gan
        self.datastore = datastore
        self.iterations = iterations
        self.quiet = quiet
        self.disc_param_save_interval = disc_param_save_interval
        self.disc_param_template = disc_param_template
        self.disc_param_save_on_error = disc_param_save_on_error
        self.quit_JDS_threshold = quit_JDS_threshold
        self.__dict__.update(kwargs)

    def pre_loop(self):
        self.learning_recorder = LearningRecorder.from_driver(self)
        self.generator_recorder = FlexGenParamRecorder.from_driver(self)
        self.discparamstats_recorder = DiscParamStatsRecorder.from_driver(self)
        self.disclearning_recorder = DiscLearningRecorder.from_driver(self)
        self.rejection_limiter = SSNRejectionLimiter.from_driver(self)
        self.disc_loss_limiter = WGANDiscLossLimiter.from_driver(self)

    def post_disc_update(self, gen_step, disc_step, Dloss, Daccuracy, SSsolve_time, gradient_time, model_info):
        self.disclearning_recorder.record(gen_step, disc_step, Dloss, Daccuracy, SSsolve_time, gradient_time,
                                          model_info.rejections, model_info.unused)
        nnorms = self.discparamstats_recorder.record(gen_step, disc_step)
        check_disc_param(self.datastore, self.gan.discriminator, nnorms)
        self.rejection_limiter(model_info.rejections)
        self.disc_loss_limiter(Dloss)

    def post_update(self, gen_step, update_result):
        self.learning_recorder.record(gen_step, update_result)
        jj, dd, ss = self.generator_recorder.record(gen_step)
        if is_at_interval(gen_step, self.disc_param_save_interval):
            param_file.dump(self.gan.discriminator,
                            self.datastore.path('disc_param', self.disc_param_template.format(gen_step)))
        if is_at_interval(gen_step, self.checkpoint_interval):
            self.gan.save_checkpoint(self.datastore.path('checkpoint.pkl'), gen_step)
        self.datastore.flush_all()
        # NB: the reference exponentiates (J, D, S) here (they used to be stored as logs) and compares with the
        # original parameters; kept for identical exit behaviour (drivers.py:147-152).
        maybe_quit(self.datastore, JDS_fake=list(map(np.exp, [jj, dd, ss])),
                   JDS_true=list(map(ssnode.DEFAULT_PARAMS.get, 'JDS')),
                   quit_JDS_threshold=self.quit_JDS_threshold)

    def iterate(self, update_func):
        if self.disc_param_save_on_error:
            inner = update_func

            def update_func(gen_step):
                param_file.dump(self.gan.discriminator, self.datastore.path('disc_param', 'pre_error.npz'))
                try:
                    return inner(gen_step)
                except Exception:
                    param_file.dump(self.gan.discriminator, self.datastore.path('disc_param', 'post_error.npz'))
                    raise
        self.pre_loop()
        logger.info('%s: start iterations', self.__class__.__name__)
        with recording_exit_reason(self.datastore):
            for gen_step in range(self.start_step, self.iterations):
                self.post_update(gen_step, update_func(gen_step))
        logger.info('%s: maximum iterations reached', self.__class__.__name__)

    def run(self, gan):
        if self.resume_from:
            self.start_step = gan.load_checkpoint(self.resume_from)
            logger.info('resumed from %s: continuing with generator step %d', self.resume_from, self.start_step)
        learning_it = gan.learning(self.start_step)
        state = {}

        def update_func(k):
            while True:
                info = next(learning_it)
                if info.is_discriminator:
                    self.post_disc_update(info.gen_step, info.disc_step, info.disc_loss, info.accuracy,
                                          info.gen_time, info.disc_time, ssnode.null_FixedPointsInfo)
                    state['disc_info'] = info
                else:
                    assert info.gen_step == k
                    disc_info = state['disc_info']
                    data_mean = _host(disc_info.xd).mean(axis=0).tolist()
                    gen_mean = _host(disc_info.xg).mean(axis=0).tolist()
                    self.datastore.tables.saverow('TC_mean.csv', gen_mean + data_mean)
                    return Namespace(info=info, disc_info=disc_info)
        self.iterate(update_func)


class BPTTcWGANDriver(BPTTWGANDriver):
    """drivers.py:340-351."""

    def post_update(self, gen_step, update_result):
        if is_at_interval(gen_step, self.tc_stats_record_interval):
            self.tuning_curve_recorder.record(gen_step, update_result.disc_info)
        super(BPTTcWGANDriver, self).post_update(gen_step, update_result)

    def pre_loop(self):
        super(BPTTcWGANDriver, self).pre_loop()
        self.tuning_curve_recorder = ConditionalTuningCurveStatsRecorder.from_driver(self)


class MomentMatchingDriver(object):
    """drivers.py:354-421."""

    def __init__(self, mmatcher, datastore, iterations, quiet, gen_moments_record_interval, quit_JDS_threshold=-1):
        self.mmatcher = mmatcher
        self.datastore = datastore
        self.iterations = iterations
        self.quiet = quiet
        self.gen_moments_record_interval = gen_moments_record_interval
        self.quit_JDS_threshold = quit_JDS_threshold

    gan = property(lambda self: self.mmatcher)        # for FlexGenParamRecorder

    def pre_loop(self):
        self.learning_recorder = MMLearningRecorder.from_driver(self)
        self.gen_moments_recorder = GenMomentsRecorder.from_driver(self)
        self.generator_recorder = FlexGenParamRecorder.from_driver(self)

    def post_update(self, gen_step, update_result):
        self.learning_recorder.record(gen_step, update_result)
        if is_at_interval(gen_step, self.gen_moments_record_interval):
            self.gen_moments_recorder.record(gen_step, update_result)
        jj, dd, ss = self.generator_recorder.record(gen_step)
        self.datastore.flush_all()
        maybe_quit(self.datastore, JDS_fake=list(map(np.exp, [jj, dd, ss])),
                   JDS_true=list(map(ssnode.DEFAULT_PARAMS.get, 'JDS')),
                   quit_JDS_threshold=self.quit_JDS_threshold)

    def iterate(self, update_func):
        self.pre_loop()
        logger.info('%s: start iterations', self.__class__.__name__)
        with recording_exit_reason(self.datastore):
            for gen_step in range(self.iterations):
                self.post_update(gen_step, update_func(gen_step))
        logger.info('%s: maximum iterations reached', self.__class__.__name__)

    def run(self, learner):
        learning_it = learner.learning()

        def update_func(k):
            info = next(learning_it)
            assert info.step == k
            return info
        self.iterate(update_func)
