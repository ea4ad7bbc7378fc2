"""Resident (plan) API: ratings, level schedule and factors stay in HBM between calls.

Python face of mfx_sgd_plan_* (include/mfx.h).  The recommender classes and bench.py use
it; the one-shot functions in mfrec_amd.lib are create+run+destroy wrappers of the same."""
import ctypes as C

import numpy as np

from . import _shim as S

LINEAR, LOGISTIC = 0, 1


class SgdPlan(object):
    SCHEDULE = {"auto": 0, "host": 1, "device": 2}
    EPOCH = {"auto": 0, "levels": 1, "ticket": 2, "static": 3}

    def __init__(self, ratings_index, ratings, nbr_users, nbr_items, dim, storage="f64", schedule="auto",
                 dot="ordered", epoch="auto"):
        """schedule: who builds the dependency-level schedule -- "host" (sequential pass + threaded
        counting sort), "device" (radix sorts + one frontier launch per level in HBM) or "auto"
        (device for >= 2^20 ratings unless the data needs more than nnz/1000 levels).  Same result.
        epoch: the kernel of a plain full-k epoch (include/mfx.h MFX_EPOCH_*): "levels" (one launch per dependency
        level), "ticket" (user rows resident, runs by ticket: input grouped by user), "static" (the smaller side's
        rows owned by fixed waves: any order, must fit the register file) or "auto" (cost model).  Same bits."""
        if epoch not in self.EPOCH:
            raise ValueError("epoch must be 'auto', 'levels', 'ticket' or 'static'")
        if schedule not in self.SCHEDULE:
            raise ValueError("schedule must be 'auto', 'host' or 'device'")
        pi, pr, nnz = S.ratings_args(ratings_index, ratings)
        self.nu, self.ni, self.k, self.nnz = int(nbr_users), int(nbr_items), int(dim), nnz
        self._h = S.lib().mfx_sgd_plan_create3(pi, pr, nnz, self.nu, self.ni, self.k, self.SCHEDULE[schedule],
                                               self.EPOCH[epoch])
        if not self._h:
            msg = S.last_error()
            raise (ValueError if ("outside" in msg or "bad argument" in msg or "MFX_EPOCH" in msg
                                  or "resident-side" in msg) else S.MfxError)(msg)
        if storage not in ("f64", "f32", "bf16"):
            raise ValueError("storage must be 'f64' (exact), 'f32' (fast mode) or 'bf16' (measured option)")
        self.storage = storage
        if storage != "f64":
            S.check(S.lib().mfx_sgd_plan_set_storage(self._h, 32 if storage == "f32" else 16))
        # dot: "ordered" = the reference's left-to-right sum (bit-exact); "butterfly" = tolerance mode
        # (wave butterfly with a fixed association + FMA: deterministic, not bit-identical)
        if dot not in ("ordered", "butterfly"):
            raise ValueError("dot must be 'ordered' (exact) or 'butterfly' (tolerance mode)")
        self.dot = dot
        if dot == "butterfly":
            S.check(S.lib().mfx_sgd_plan_set_dot(self._h, 1))

    def close(self):
        if getattr(self, "_h", None):
            S.lib().mfx_sgd_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: the library may already be gone
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        a = (C.c_int64 * 8)()
        S.check(S.lib().mfx_sgd_plan_info(self._h, a))
        keys = ("n_levels", "max_level", "launches_per_epoch", "tail_launches", "ratings_per_wave",
                "device_bytes", "schedule_us", "funk_launches_per_epoch")
        d = dict(zip(keys, [int(x) for x in a]))
        d["schedule_builder"] = {1: "host", 2: "device"}.get(S.lib().mfx_sgd_plan_schedule_builder(self._h), "?")
        b = (C.c_int64 * 4)()
        S.check(S.lib().mfx_sgd_plan_res_info(self._h, b))
        names = {1: "levels", 2: "ticket", 3: "static"}
        d["epoch_kernel"], d["epoch_kernel_prepared"] = names.get(int(b[0]), "?"), names.get(int(b[1]), "?")
        return d

    def kernel_name(self, mode=LINEAR):
        """The k1_level instantiation the library launches for this plan (from the library itself)."""
        buf = C.create_string_buffer(128)
        S.check(S.lib().mfx_sgd_plan_kernel_name(self._h, int(mode), buf, 128))
        return buf.value.decode()

    def set_factors(self, u, v, items_bias=None, users_bias=None):
        pu, pv, nu, ni = S.factor_args(u, v, self.k)
        if (nu, ni) != (self.nu, self.ni):
            raise ValueError("factor shapes do not match the plan")
        pib = S.f64(items_bias, "items_bias", 1, (ni,), allow_none=True)
        pub = S.f64(users_bias, "users_bias", 1, (nu,), allow_none=True)
        S.check(S.lib().mfx_sgd_set_factors(self._h, pu, pv, pib, pub))

    def get_factors(self, u, v, items_bias=None, users_bias=None):
        pu, pv, nu, ni = S.factor_args(u, v, self.k)
        if (nu, ni) != (self.nu, self.ni):
            raise ValueError("factor shapes do not match the plan")
        pib = S.f64(items_bias, "items_bias", 1, (ni,), allow_none=True)
        pub = S.f64(users_bias, "users_bias", 1, (nu,), allow_none=True)
        S.check(S.lib().mfx_sgd_get_factors(self._h, pu, pv, pib, pub))

    def run_kmf(self, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, update_users=1,
                update_items=1, want_rmse=True):
        rm = np.zeros(max(int(nbr_epochs), 1))
        S.check(S.lib().mfx_sgd_run_kmf(self._h, int(mode), int(nbr_epochs), float(learning_rate),
                                        float(K_users), float(K_items), float(K_bias),
                                        int(bool(update_users)), int(bool(update_items)),
                                        S.f64(rm, "rmse", 1) if want_rmse else None))
        return rm[:max(int(nbr_epochs), 0)]

    def profile_kmf(self, mode, learning_rate, K_users, K_items, K_bias):
        """One instrumented epoch: (sum of kernel durations in ms, number of launches)."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        S.check(S.lib().mfx_sgd_profile_kmf(self._h, int(mode), float(learning_rate),
                                            float(K_users), float(K_items), float(K_bias),
                                            C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def run_funk(self, min_epochs, min_improvement, f_init, learning_rate, K, overall_avg=1.0,
                 use_bias=False, update_users=1, update_items=1, epoch_cap=0):
        hist = np.zeros(max(1, int(min_epochs)) * self.k * 4 + 1024)
        n = C.c_int64(0)
        S.check(S.lib().mfx_sgd_run_funk(self._h, int(min_epochs), float(min_improvement),
                                         float(f_init), float(learning_rate), float(K),
                                         float(overall_avg), int(bool(use_bias)),
                                         int(bool(update_users)), int(bool(update_items)),
                                         int(epoch_cap), S.f64(hist, "hist", 1), hist.shape[0],
                                         C.byref(n)))
        return hist[:min(n.value, hist.shape[0])]

    def run_learned_bias(self, min_epochs, min_improvement, f_init, learning_rate,
                         learning_rate_users, learning_rate_items, K_feature, K_bias, overall_avg,
                         epoch_cap=0):
        hist = np.zeros(max(1, int(min_epochs)) * self.k * 4 + 1024)
        n = C.c_int64(0)
        S.check(S.lib().mfx_sgd_run_learned_bias(
            self._h, int(min_epochs), float(min_improvement), float(f_init), float(learning_rate),
            float(learning_rate_users), float(learning_rate_items), float(K_feature), float(K_bias),
            float(overall_avg), int(epoch_cap), S.f64(hist, "hist", 1), hist.shape[0], C.byref(n)))
        return hist[:min(n.value, hist.shape[0])]

    def run_funk_sharded(self, min_epochs, min_improvement, f_init, learning_rate, K, scale,
                         overall_avg=1.0, use_bias=False, epoch_cap=0):
        hist = np.zeros(max(1, int(min_epochs)) * self.k * 4 + 1024)
        n = C.c_int64(0)
        S.check(S.lib().mfx_sgd_run_funk_sharded(
            self._h, int(min_epochs), float(min_improvement), float(f_init), float(learning_rate),
            float(K), float(overall_avg), int(bool(use_bias)), int(epoch_cap), float(scale),
            S.f64(hist, "hist", 1), hist.shape[0], C.byref(n)))
        return hist[:min(n.value, hist.shape[0])]

    def run_kmf_sharded(self, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale):
        rm, se = np.zeros(max(int(nbr_epochs), 1)), np.zeros(max(int(nbr_epochs), 1))
        S.check(S.lib().mfx_sgd_run_kmf_sharded(self._h, int(mode), int(nbr_epochs),
                                                float(learning_rate), float(K_users),
                                                float(K_items), float(K_bias), float(scale),
                                                S.f64(rm, "rmse", 1), S.f64(se, "se", 1)))
        return rm[:int(nbr_epochs)], se[:int(nbr_epochs)]


def run_kmf_fake_world(plans, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale):
    """All shards of the user-sharded scheme on ONE GPU, rank-ordered on-device sum in place of the
    all-reduce (mfx_sgd_run_kmf_fake_world).  -> se f64[world, nbr_epochs]."""
    world = len(plans)
    arr = (C.c_void_p * world)(*[p._h for p in plans])
    se = np.zeros((world, max(int(nbr_epochs), 1)))
    S.check(S.lib().mfx_sgd_run_kmf_fake_world(arr, world, int(mode), int(nbr_epochs), float(learning_rate),
                                               float(K_users), float(K_items), float(K_bias), float(scale),
                                               S.f64(se, "se", 2)))
    return se[:, :int(nbr_epochs)]


class Timer(object):
    """HIP-event stopwatch on the library's stream."""
    def __enter__(self):
        S.check(S.lib().mfx_timer_start())
        return self

    def __exit__(self, *a):
        ms = C.c_double(0.0)
        S.check(S.lib().mfx_timer_stop(C.byref(ms)))
        self.ms = ms.value
