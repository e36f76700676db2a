#!/usr/bin/env python3
"""CLI mirror of the reference's run_train.py (flags of run_train.py:32-47) on the MI355X-native engine.
Multi-GPU: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 run_train.py ...`."""
import argparse
import os
import numpy as np
import torch

if __name__ == "__main__":
    np.random.seed(218); torch.manual_seed(218)                      # run_train.py:20-21
    p = argparse.ArgumentParser(description="trains SignalTrain network", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--apex', help="mixed precision optimization level as in the reference: O0 = fp32; O1/O2/O3 = float16 operands, "
                   "fp32 accumulation, loss scaling, clip over all parameters (the library's f16_all arithmetic; there is no Apex on ROCm)",
                   default="O0", choices=["O0", "O1", "O2", "O3"])
    p.add_argument('--dtype', help="arithmetic of the accelerated step, overrides --apex: f32 | f32x3 (fp32-grade GEMMs on the bf16 matrix pipe) | bf16 | bf16_all | f16 | f16_all",
                   default=None, choices=["f32", "f32x3", "bf16", "bf16_all", "f16", "f16_all"])
    p.add_argument('--gpus', type=int, default=None, help="data-parallel world size this job is meant to run on; launch with "
                   "`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 run_train.py --gpus N ...` "
                   "(one process per GPU); checked against WORLD_SIZE")
    p.add_argument('--device-feed', action='store_true', default=True, help="(default) generate every training minibatch on the GPU (st_synth_comp4c) and keep the validation set in HBM")
    p.add_argument('--host-feed', dest='device_feed', action='store_false', help="the reference's feed instead: a torch DataLoader with 10 CPU workers over the Dataset items")
    p.add_argument('--resume-optimizer', action='store_true', help="restore Adam's moments (and, if the checkpoint belongs to this schedule, the position in the run) from --checkpoint")
    p.add_argument('-b', '--batch', type=int, help="batch size (per GPU); the reference's default.  On MI355X multiples of 256 windows (8192-sample window) fill the tile rounds of the analysis "
                                                        "forward: 379 k windows/s at 200, 418-426 k at 256, 477 k at 512 in fp32 (DESIGN.md section 5)", default=200)
    p.add_argument('--checkpoint', help='name of checkpoint .tar file to start from', default='modelcheckpoint.tar')
    p.add_argument('-c', '--compand', help='mu-law compand the audio of a file dataset (datasets.py:218-220)', action='store_true')
    p.add_argument('--effect', help="effect to learn, the reference's keys (run_train.py:55-80): comp_4c | comp_large | files are built here "
                   "(comp_4c_large is kept as an alias of comp_large); the reference's other keys are named in the error message", default='comp_4c')
    p.add_argument('--epochs', type=int, default=1000)
    p.add_argument('--lrmax', type=float, help="maximum learning rate", default=1e-4)
    p.add_argument('-n', '--num', type=int, help='number of data points per epoch', default=200000)
    p.add_argument('--path', help='dataset directory with Train/, Val/ wav pairs and effect_info.ini (use with --effect files)', default=None)
    p.add_argument('--sr', type=int, default=44100)
    p.add_argument('--scale', type=float, help='scale factor (of input size & whole model)', default=1.0)
    p.add_argument('--shrink', type=int, help='shrink output chunk relative to input by this divisor', default=4)
    p.add_argument('-t', '--target', help='accepted for compatibility', default="stream")
    args = p.parse_args()
    # the reference's effect table (run_train.py:55-80); the hot path carries the 4-knob compressor family and file pairs
    EFFECTS = {'comp_4c': 'Compressor_4c', 'comp_large': 'Compressor_4c_Large', 'comp_4c_large': 'Compressor_4c_Large', 'files': 'FileEffect'}
    NOT_BUILT = ('comp', 'comp_t', 'comp_one', 'denoise', 'lowpass')      # audio.Compressor / Comp_Just_Thresh / Compressor_4c_OneSetting / Denoise / LowPass
    if args.effect not in EFFECTS:
        if args.effect in NOT_BUILT or 'VST' in args.effect:
            raise SystemExit(f"--effect {args.effect}: a key of the reference's run_train.py that signaltrain_amd does not build (its scope is the comp_4c training path: "
                             f"{', '.join(k for k in EFFECTS if k != 'comp_4c_large')}); not built: {', '.join(NOT_BUILT)}, VST*")
        raise SystemExit(f"Effect option '{args.effect}' is not yet added (available: {', '.join(k for k in EFFECTS if k != 'comp_4c_large')})")       # run_train.py:79-80
    if args.target not in ("chunk", "stream"):
        raise SystemExit(f"Error, invalid target type: {args.target}")                # run_train.py:90-92

    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one process per GPU with\n  python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29500 run_train.py --gpus {args.gpus} ...")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # bootstrap channel only (RCCL unique id, logging): the gradient exchange runs on the library's own RCCL communicator
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")              # one node: do not depend on the host name resolving
        dist.init_process_group("gloo")
    from signaltrain_amd import audio, train
    effect = audio.FileEffect(args.path, sr=args.sr) if args.effect == 'files' else getattr(audio, EFFECTS[args.effect])()
    train.train(epochs=args.epochs, n_data_points=args.num, batch_size=args.batch, device=torch.device("cuda", local),
                effect=effect, datapath=(args.path if args.effect == 'files' else None), sr=args.sr, scale_factor=args.scale, shrink_factor=args.shrink, apex_opt=args.apex,
                target_type=args.target, lr_max=args.lrmax, in_checkpointname=args.checkpoint, compand=args.compand,
                compute_dtype=args.dtype, device_feed=args.device_feed, resume_optimizer=args.resume_optimizer)
    if dist.is_initialized():
        dist.destroy_process_group()
