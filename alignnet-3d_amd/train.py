#!/usr/bin/env python3
"""Driver of the drop-in: same command line, config handling, epoch loops, checkpoints cadence and output
files as the reference's train.py, with the TF graph/session replaced by the MI355X engine.

    python train.py {train,eval_only} --config configs/X.json [--eval_epoch E] [--use_old_results]
                    [--refineICP] [--its K] [--refineICPmethod p2p]

Launch under `python -m torch.distributed.run --nproc-per-node N` for data-parallel training / batch-split
evaluation (one process per GPU; gradient all-reduce on RCCL inside the engine)."""
import argparse
import copy
import datetime
import logging
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import provider  # noqa: E402
import evaluation  # noqa: E402
import models.tp8 as MODEL  # noqa: E402
from config import load_config, save_config, configGlobal as cfg  # noqa: E402
from alignnet3d import parallel  # noqa: E402

logger = logging.getLogger("tp")
CKPT_EXT = ".aln3"   # own container format; TF tensor-bundle import is a later row of SURVEY 8(f)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("operation", choices=["train", "eval_only"], help="Operation to run")
    p.add_argument("--config", required=True, default="", help="Config file")
    p.add_argument("--refineICP", action="store_true", help="Whether the results should be refined with ICP")
    p.add_argument("--its", required=False, default=30, help="How many iteration the result should be refined with ICP")
    p.add_argument("--use_old_results", action="store_true", help="Use stored predictions instead of running the model")
    p.add_argument("--refineICPmethod", required=False, default="p2p", choices=["p2p"], help="ICP method for refinement")
    p.add_argument("--eval_epoch", required=False, default="199", help="Epoch to eval in eval_only mode")
    return p.parse_args(argv)


def setup_logging(logdir, rank):
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(asctime)s %(name)-12s %(levelname)-8s %(message)s", "%Y-%m-%d %H:%M:%S")
    if rank == 0:
        sh = logging.StreamHandler()
        sh.setLevel(logging.INFO)
        sh.setFormatter(fmt)
        logger.addHandler(sh)
        logfile = "%s/out.log" % logdir
        if os.path.exists(logfile):
            logfile = "%s_%s.log" % (logfile[:-4], datetime.datetime.today().strftime("%Y-%m-%d_%H-%M-%S"))
        fh = logging.FileHandler(logfile)
        fh.setLevel(logging.DEBUG)
        fh.setFormatter(fmt)
        logger.addHandler(fh)


class Run:
    def __init__(self, flags):
        self.flags = flags
        self.rank, self.local_rank, self.world = parallel.world_info()
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world)
            self.dist = dist
        import alignnet3d
        if self.world > 1 and cfg.training.batch_size % self.world:
            # the engine averages the ranks' gradients with a fixed 1 / world: unequal shards would be weighted equally
            raise SystemExit("training.batch_size (%d) must be a multiple of the number of ranks (%d)" % (cfg.training.batch_size, self.world))
        self.engine = alignnet3d.Engine(cfg, device=self.local_rank if self.world > 1 else None)
        if self.world > 1:
            self.engine.set_option("dropout_stream", self.rank)   # same initialisation on every rank (seed 0), different dropout masks
        MODEL.bind_engine(self.engine)
        # optional, not a reference key: "training": {"matmul_dtype": "bf16"} (or ALIGNNET_TRAIN_BF16=1) runs the widest conv of
        # every backbone on bf16 MFMA during training (engine option train_matmul_bf16; evaluation stays fp32)
        if os.environ.get("ALIGNNET_INFER_BF16X3", "") not in ("", "0"):
            self.engine.set_option("infer_matmul_bf16x3", 1)   # eval-mode backbone on split-bf16 products (1e-4 parity bar kept)
            logger.info("evaluation with the split-bf16 backbone")
        if str(getattr(cfg.training, "matmul_dtype", "f32")).lower() == "bf16" or os.environ.get("ALIGNNET_TRAIN_BF16", "") not in ("", "0"):
            self.engine.set_option("train_matmul_bf16", 1)
            logger.info("training with bf16 operands in the 128->C3 lifts")
        if self.dist is not None:
            parallel.init_comm(self.engine, self.dist)
            # optional, not reference keys: "training": {"sync_bn": true, "global_loss": true} (or ALIGNNET_SYNC_BN=1 / ALIGNNET_GLOBAL_LOSS=1):
            # batch statistics and the loss over the GLOBAL batch -- the reference's single-device step at batch_size;
            # default: every rank normalises with / takes the loss of its own shard
            if bool(getattr(cfg.training, "sync_bn", False)) or os.environ.get("ALIGNNET_SYNC_BN", "") not in ("", "0"):
                self.engine.set_option("sync_bn", 1)
                logger.info("BatchNorm statistics over all %d ranks (sync_bn)" % self.world)
            if bool(getattr(cfg.training, "global_loss", False)) or os.environ.get("ALIGNNET_GLOBAL_LOSS", "") not in ("", "0"):
                self.engine.set_option("global_loss", 1)
                logger.info("loss over the global batch (global_loss)")
        self.device_data = None
        if os.environ.get("ALIGNNET_DEVICE_DATASET", "") not in ("", "0"):
            # whole dataset resident in HBM, batches resampled + jittered on the device (alignnet_dataset_*): same
            # distribution as provider.load_batch + jitter_point_cloud, but the engine's own random stream, not np.random's
            self.device_data = provider.use_packed_cache()
            self.device_data.upload(self.engine)
        elif os.environ.get("ALIGNNET_PACKED_CACHE", "") not in ("", "0"):
            provider.use_packed_cache()   # bit-identical batches, no per-example file opens (alignnet3d/packed.py)
        self.icp_data = self.device_data
        if flags.refineICP and self.icp_data is None:
            # ICP refines on the FULL clouds (train.py:469), which live in HBM next to the network (alignnet_icp_refine_dataset)
            self.icp_data = provider.use_packed_cache()
            self.icp_data.upload(self.engine)
        self.train_idx = provider.getDataFiles("%s/split/train.txt" % cfg.data.basepath)
        self.val_idx = provider.getDataFiles("%s/split/val.txt" % cfg.data.basepath)
        self.batches_per_epoch = len(self.train_idx) // cfg.training.batch_size

    # ---- checkpoints (train.py:245-293, 313-322 of the reference) --------------------------------------------
    def ckpt(self, stem):
        return os.path.join(cfg.logging.logdir, stem + CKPT_EXT)

    def restore(self, base, skip_step=False):
        """Load `<base>.aln3` (own format) or, failing that, the TensorFlow tensor bundle `<base>.index` +
        `<base>.data-00000-of-00001` written by the reference's tf.train.Saver (alignnet3d/tf_bundle.py)."""
        if os.path.isfile(base + CKPT_EXT):
            self.engine.load(base + CKPT_EXT, skip_step=skip_step)
            return True
        if os.path.isfile(base + ".index"):
            from alignnet3d import tf_bundle
            tf_bundle.load_into_engine(self.engine, base, load_step=not skip_step)
            logger.info("Restored TensorFlow checkpoint %s" % base)
            return True
        return False

    def restore_for_training(self):
        if os.path.isfile(self.ckpt("model.ckpt")) or os.path.isfile(os.path.join(cfg.logging.logdir, "model.ckpt.index")):
            assert self.restore(os.path.join(cfg.logging.logdir, "model.ckpt"))
            step = self.engine.state()["step"]
            assert step % self.batches_per_epoch == 0
            logger.info("Continuing training at epoch %d" % (step // self.batches_per_epoch))
            return step // self.batches_per_epoch
        pre = cfg.training.pretraining.model
        if pre != "":
            base = pre[: -len(CKPT_EXT)] if pre.endswith(CKPT_EXT) else pre
            assert self.restore(base, skip_step=True), base   # every variable except `batch` (train.py:278-281)
            assert self.engine.state()["step"] == 0
            logger.info("Pre-trained weights loaded from %s, starting initial evaluation" % pre)
            self.eval_one_epoch("pretr", eval_only=False, do_timings=False)
            logger.info("Initial evaluation finished")
        return 0

    # ---- train.py:335-383 -------------------------------------------------------------------------------------
    def train_one_epoch(self, epoch):
        B = cfg.training.batch_size
        idxs = copy.deepcopy(self.train_idx)
        np.random.shuffle(idxs)                      # train.py:338 (np.random's stream, single process: as the reference)
        if self.dist is not None:
            # one permutation for all ranks (rank 0's): each global batch is then a partition of B distinct examples and an
            # epoch is one pass over the training set, as in the reference; every rank loads only its own slice
            idxs = parallel.broadcast_object(self.dist, idxs)
        loss_sum = 0.0
        lo, hi = parallel.shard_range(B, self.rank, self.world)
        for b in range(len(idxs) // B):
            mine = idxs[b * B + lo:b * B + hi]
            if self.device_data is not None:
                rows = self.device_data.rows_of(mine)
                res = self.engine.train_step_rows(rows, seed=int(np.random.randint(0, 2 ** 62)))   # jitter 0.01 / 0.05 (provider.py:60)
                loss_sum += res["loss"]
                continue
            batch = provider.load_batch(mine, override_batch_size=hi - lo)
            pcs1 = provider.jitter_point_cloud(batch[0])
            pcs2 = provider.jitter_point_cloud(batch[1])
            labels = dict(zip(("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles"), batch[2:]))
            res = self.engine.train_step(pcs1, pcs2, labels)
            loss_sum += res["loss"]
        n = max(len(idxs) // B, 1)
        if self.dist is not None:
            # local-BN data parallelism: every rank's EMA shadows saw only its shard's statistics; average them so
            # that all ranks evaluate (and rank 0 checkpoints) the same model, and report the mean of the ranks' shard losses
            parallel.average_ema_shadows(self.engine, self.dist)
            loss_sum = parallel.mean_scalar(self.dist, loss_sum)
        logger.info("train mean loss: %f" % (loss_sum / float(n)))

    # ---- train.py:386-545 -------------------------------------------------------------------------------------
    def eval_one_epoch(self, epoch, eval_only, do_timings, override_batch_size=None):
        flags = self.flags
        B = cfg.training.batch_size if override_batch_size is None else override_batch_size
        val = self.val_idx
        nval = len(val)
        eval_dir = "%s/val/eval%s" % (cfg.logging.logdir, str(epoch).zfill(6))
        base_eval_dir = eval_dir
        refine = bool(eval_only and flags.refineICP)
        if flags.refineICP:   # train.py:401-402 (the suffix compares the raw flag with the int 30, as the reference does)
            eval_dir = "%s/refined_%s%s" % (eval_dir, flags.refineICPmethod, ("_" + str(flags.its)) if flags.its != 30 else "")
        old = None
        if flags.use_old_results:   # train.py:423-426
            old = {k: np.load("%s/%s.npy" % (base_eval_dir, k)) for k in ("pred_translations", "pred_angles", "pred_s2_pc1centers")}
        if self.rank == 0:
            if os.path.isdir(eval_dir):
                backup, k = "%s_backup_%d" % (eval_dir, int(time.time())), 0
                while os.path.exists(backup):   # the reference's second-resolution name collides when an eval takes < 1 s
                    k += 1
                    backup = "%s_backup_%d_%d" % (eval_dir, int(time.time()), k)
                os.rename(eval_dir, backup)
            os.makedirs(eval_dir, exist_ok=True)
        names3 = ("pred_translations", "pred_s1_pc1centers", "pred_s1_pc2centers", "pred_s2_pc1centers", "pred_s2_pc2centers")
        store = {k: np.empty((nval, 3), np.float32) for k in names3}
        for k in ("pred_angles", "pred_s2_pc1angles", "pred_s2_pc2angles"):
            store[k] = np.empty((nval, 1), np.float32)
        gt_t, gt_a, gt_c1 = np.empty((nval, 3), np.float32), np.empty((nval, 1), np.float32), np.empty((nval, 3), np.float32)
        loss_sum, cumulated, full_batches = 0.0, 0.0, nval // B
        for b in range(int(np.ceil(nval / B))):
            s, e = b * B, min((b + 1) * B, nval)
            n = e - s
            batch = provider.load_batch(val[s:e], override_batch_size=override_batch_size, dont_load_pointclouds=self.device_data is not None)
            lo, hi = parallel.shard_range(n, self.rank, self.world)
            t0 = time.time()
            if hi > lo and self.device_data is not None:
                ep = self.engine.forward_rows(self.device_data.rows_of(val[s:e])[lo:hi], seed=int(np.random.randint(0, 2 ** 62)))
            elif hi > lo:
                ep = self.engine.forward(batch[0][lo:hi], batch[1][lo:hi])   # any batch size: no padding rows needed
            else:
                ep = {k: np.zeros((0, 3 if "logits" not in k else 2 * cfg.model.angles.num_bins), np.float32) for k in
                      ("pred_translations", "pred_remaining_angle_logits", "pred_s1_pc1centers", "pred_s1_pc2centers",
                       "pred_s2_pc1centers", "pred_s2_pc2centers", "pred_pc1angle_logits", "pred_pc2angle_logits")}
            cumulated += time.time() - t0
            if self.world == 1 and n == B:   # last (partial) batch is not counted (train.py:458)
                labels = dict(zip(("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles"),
                                  [a[:n] for a in batch[2:]]))
                loss_sum += self.engine.eval_loss(labels, n)[0]
            # world > 1: the loss couples the rows of a batch ([B,B] terms, tp8.py:279,327), so a shard's loss is not the batch's;
            # it is not computed and is logged as n/a below rather than as a number that looks real
            if self.dist is not None:
                counts = [parallel.shard_range(n, r, self.world) for r in range(self.world)]
                counts = [c[1] - c[0] for c in counts]
                ep = {k: parallel.gather_rows(self.dist, v, counts) for k, v in ep.items()}
            a1 = MODEL.classLogits2angle(ep["pred_pc1angle_logits"])
            a2 = MODEL.classLogits2angle(ep["pred_pc2angle_logits"])
            ar = MODEL.classLogits2angle(ep["pred_remaining_angle_logits"])
            pred_angles = a2 - a1 + ar                           # train.py:456
            if refine and self.rank == 0:
                # train.py:463-481: ICP on the full clouds seeded by the prediction; the refined transform is about the
                # origin, so the rotation centre becomes 0 and the angle is the z Euler angle of its rotation part
                src = old if old is not None else None
                pt = src["pred_translations"][s:e] if src else ep["pred_translations"]
                pa = src["pred_angles"][s:e, 0] if src else pred_angles
                pc = src["pred_s2_pc1centers"][s:e] if src else ep["pred_s2_pc1centers"]
                inits = [evaluation.get_mat_angle(pt[i], pa[i], rotation_center=pc[i]) for i in range(n)]
                t0 = time.time()
                T = self.engine.icp_refine_rows(self.icp_data.rows_of(val[s:e]), inits, radius=0.1, its=int(flags.its))["transforms"]
                cumulated += time.time() - t0
                ep["pred_translations"] = T[:, :3, 3].astype(np.float32)
                pred_angles = np.arctan2(T[:, 1, 0], T[:, 0, 0])
                ep["pred_s2_pc1centers"] = np.zeros((n, 3), np.float32)
            store["pred_angles"][s:e, 0] = pred_angles
            store["pred_s2_pc1angles"][s:e, 0], store["pred_s2_pc2angles"][s:e, 0] = a1, a2
            for k in names3:
                store[k][s:e] = ep[k]
            gt_t[s:e], gt_a[s:e], gt_c1[s:e] = batch[2][:n], batch[3][:n], batch[4][:n]
        mean_loss = loss_sum / full_batches if full_batches > 0 else 0.0
        mean_time = cumulated / float(nval)
        if do_timings:
            print("Timing bs=%s: %s" % (override_batch_size, mean_time))
        elif self.rank == 0:
            for inv in (False, True):
                ev = evaluation.evaluate(cfg, val, store["pred_translations"], store["pred_angles"], gt_t, gt_a,
                                         store["pred_s2_pc1centers"], gt_c1, eval_dir=eval_dir, accept_inverted_angle=inv,
                                         mean_time=mean_time)
                pct = lambda v: " ".join("%.2f%%" % (a * 100.0) for a in v)
                logger.info("Mean translation distance: %s, Mean angle distance: %s, Levels: %s, Translation levels: %s, "
                            "Angle levels: %s, Mean ex. time: %.5f" % (ev.mean_dist_translation, ev.mean_dist_angle, pct(ev.corr_levels),
                                                                       pct(ev.corr_levels_translation), pct(ev.corr_levels_angles), mean_time))
        if self.rank == 0:
            for k, v in store.items():
                np.save("%s/%s.npy" % (eval_dir, k), v)
            logger.info("val mean loss: %f" % mean_loss if self.world == 1 else "val mean loss: n/a (batch split over %d ranks)" % self.world)
        return mean_time

    # ---- train.py:187-332 -------------------------------------------------------------------------------------
    def train(self, eval_only=False, eval_epoch=None, eval_only_model_to_load=None, do_timings=False, override_batch_size=None):
        start_epoch = 0
        if eval_only:
            model_to_load = cfg.logging.logdir if eval_only_model_to_load is None else eval_only_model_to_load   # train.py:247-249
            if not self.flags.use_old_results and not do_timings:
                base = os.path.join(model_to_load, "model-%s" % eval_epoch)
                assert self.restore(base), base + "{.aln3,.index}"
                if eval_only_model_to_load is None:   # a held-out model was trained on another split: no epoch check (train.py:255)
                    step = self.engine.state()["step"]
                    assert step % self.batches_per_epoch == 0
                    assert step // self.batches_per_epoch - 1 == int(eval_epoch)
            start_epoch = int(eval_epoch)
            logger.info("Evaluating at epoch %d" % start_epoch)
        else:
            start_epoch = self.restore_for_training()
        start = time.time()
        try:
            for epoch in range(start_epoch, cfg.training.num_epochs):
                st = self.engine.state()
                logger.info("**** EPOCH %03d ****    lr: %.8f, bn_decay: %.8f" % (epoch, st["learning_rate"], st["bn_decay"]))
                if not eval_only:
                    self.train_one_epoch(epoch)
                if do_timings:
                    for _ in range(10):
                        self.eval_one_epoch(epoch, eval_only, True, override_batch_size)
                else:
                    self.eval_one_epoch(epoch, eval_only, False)
                if eval_only:
                    break
                last = epoch == cfg.training.num_epochs - 1
                if self.rank == 0 and (epoch % 2 == 0 or last):
                    self.engine.save(self.ckpt("model.ckpt"))
                    logger.info("Model saved in file: %s" % self.ckpt("model.ckpt"))
                if self.rank == 0 and (epoch % 5 == 0 or last or cfg.evaluation.save_every_epoch):
                    self.engine.save(self.ckpt("model-%d" % epoch))
                    logger.info("Model saved in file: %s" % self.ckpt("model-%d" % epoch))
                el = time.time() - start
                logger.info("Finished epoch %d. Time elapsed: %s, Time remaining: %s" % (
                    epoch, datetime.timedelta(seconds=el), datetime.timedelta(seconds=el / (epoch + 1) * (cfg.training.num_epochs - epoch - 1))))
            logger.info("Finished Training")
        except KeyboardInterrupt:
            logger.info("Interrupted")


def main(argv=None):
    flags = parse_args(argv)
    load_config(flags.config)
    rank = parallel.world_info()[0]
    os.makedirs(cfg.logging.logdir, exist_ok=True)
    if rank == 0:
        copyfile = "%s/config.json" % cfg.logging.logdir
        if os.path.exists(copyfile):
            copyfile = "%s_%s.json" % (copyfile[:-5], datetime.datetime.today().strftime("%Y-%m-%d_%H-%M-%S"))
        save_config(copyfile)
    assert cfg.model.model == "tp8"
    setup_logging(cfg.logging.logdir, rank)
    logger.debug(cfg)
    if cfg.evaluation.has("special"):
        mode = cfg.evaluation.special.mode
        if mode == "timings":        # train.py:555-559: bs = 32, ten repeated eval epochs, no checkpoint needed
            for bs in [32]:
                cfg.training.batch_size = bs
                Run(flags).train(eval_only=True, eval_epoch=flags.eval_epoch, do_timings=True, override_batch_size=bs)
        elif mode == "held":         # train.py:553-554: eval_only with the checkpoint of another run's logdir
            Run(flags).train(eval_only=True, eval_epoch=flags.eval_epoch, eval_only_model_to_load=cfg.evaluation.special.held.model)
        elif mode == "icp":
            raise NotImplementedError("evaluation.special.mode=icp runs the reference's CPU ICP baselines (icp.py:80-330: global registration, "
                                      "Go-ICP): out of scope (SURVEY 2); --refineICP is the supported ICP path")
        else:
            assert False
    elif flags.operation == "train":
        Run(flags).train()
    else:
        Run(flags).train(eval_only=True, eval_epoch=flags.eval_epoch)


if __name__ == "__main__":
    main()
