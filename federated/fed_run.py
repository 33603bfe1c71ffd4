#!/usr/bin/env python3
"""Drop-in for the reference's federated/fed_run.py (fedavg / fedbn / deepall paths): same flags (:458-503), console lines,
checkpoint dicts ({'server_model', 'a_iter'}, best and '_latest', :733-766) and --resume / --test.

Two launch forms
  * python federated/fed_run.py ...                      one process, clients trained one after another on
                                                         one GPU, in-process communication() (the reference's
                                                         schedule, fed_run.py:663-684)
  * torchrun --nproc-per-node K federated/fed_run.py ... one process per GPU = one client per GPU (K = number
                                                         of --source domains); the FedAvg average is ONE RCCL
                                                         all-reduce of the flat fp32 state per global round.
--mode fedbn (fed_run.py:388-399, :693-698, :735-759): clients keep every state entry whose key contains 'bn',
are validated with their local model, and checkpoints also carry 'model_{k}' state dicts.
Out of scope: adafea / fedprox aggregation variants, RSC / Jigsaw / MixStyle /
FedDG, Tent, tensorboard / Excel logging."""
import argparse
import copy
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ccst_amd import data, fed  # noqa: E402
from ccst_amd.nets.models import get_network, nets_map  # noqa: E402

available_datasets = ["art_painting", "cartoon", "photo", "sketch", 'art', 'clipart', 'product', 'real_world',
                      'MNIST', 'MNIST_M', 'SVHN', 'SynthDigits', 'USPS',
                      'hospital1', 'hospital2', 'hospital3', 'hospital4', 'hospital5']


def parse():
    parser = argparse.ArgumentParser()
    parser.add_argument('--log', action='store_true', help='whether to make a log')
    parser.add_argument('--test', action='store_true', help='test the pretrained model')
    parser.add_argument('--tent_test', action='store_true', help='test the pretrained model with the Tent test-time optimization')
    parser.add_argument('--tent_test_on-the-fly', action='store_true', help='test the pretrained model with the Tent test-time optimization one by one')
    parser.add_argument('--IN_test', action='store_true', help='test the pretrained model using IN with affine')
    parser.add_argument('--batch', type=int, default=32, help='batch size')
    parser.add_argument('--iters', type=int, default=500, help='iterations for communication')
    parser.add_argument('--wk_iters', type=int, default=1, help='optimization iters in local worker between communication')
    parser.add_argument('--mode', type=str, default='fedavg', choices=['fedavg', 'fedbn', 'adafea', 'fedprox', 'deepall'])
    parser.add_argument('--mu', type=float, default=1e-2)
    parser.add_argument('--save_path', type=str, default='../checkpoint', help='path to save the checkpoint')
    parser.add_argument('--resume', action='store_true', help='resume training from the save path checkpoint')
    parser.add_argument('--percent', type=float, default=0.1)
    parser.add_argument("--n_classes", "-c", type=int, default=10, help="Number of classes")
    parser.add_argument("--dataset", choices=['pacs', 'officehome', 'digitsfive', 'camelyon17'], default='pacs')
    parser.add_argument("--source", choices=available_datasets, help="Source", nargs='+')
    parser.add_argument("--target", choices=available_datasets, help="Target")
    parser.add_argument("--limit_source", default=None, type=int)
    parser.add_argument("--limit_target", default=None, type=int)
    parser.add_argument("--val_size", type=float, default="0.1")
    parser.add_argument('--lr', type=float, default=1e-2, help='learning rate')
    parser.add_argument("--fusion_mode", default='no_fusion',
                        choices=['no_fusion'] + ['adain-%s-K%d' % (m, k) for m in ('single', 'overall') for k in (1, 2, 3, 4)])
    parser.add_argument("--dg_method", choices=['no_DG', 'RSC', 'Jigsaw', 'MixStyle', 'feddg'], default='no_DG')
    parser.add_argument("--network", choices=nets_map.keys(), default="resnet50")
    parser.add_argument("--image_size", type=int, default=225, help="Image size")
    parser.add_argument("--min_scale", default=0.8, type=float)
    parser.add_argument("--max_scale", default=1.0, type=float)
    parser.add_argument("--random_horiz_flip", default=0.0, type=float)
    parser.add_argument("--tf_logger", type=bool, default=True)
    parser.add_argument('--gpu', type=int, default=0, help='gpu device number')
    parser.add_argument('--seed', type=int, default=1, help='random seed number')
    parser.add_argument('--save_freq', type=int, default=1)
    # Jigsaw / FedDG hyper-parameters (fed_run.py:498-502): parsed with the reference's defaults; the methods that read them
    # (--dg_method Jigsaw / feddg) are outside the hot path and refuse in main()
    parser.add_argument("--bias_whole_image", default=0.9, type=float, help="If set, will bias the training procedure to show more often the whole image")
    parser.add_argument("--jig_weight", type=float, default=0.7, help="Weight for the jigsaw puzzle loss")
    parser.add_argument('--meta_step_size', type=float, default=1e-3, help='meta learning rate')
    parser.add_argument('--clip_value', type=float, default=1.0, help='gradient clip')
    # additions
    parser.add_argument('--synthetic', type=int, default=0, help='N seeded synthetic images per client instead of the list files')
    parser.add_argument('--txt_root', type=str, default='data/txt_lists')
    parser.add_argument('--hip_graph', nargs='?', const=True, default=False,
                        help='capture the train iteration into a HIP graph and replay it per batch (launch-bound models: ResNet18, small batches); '
                             '"--hip_graph auto": time both loops during the first epoch and keep the faster')
    parser.add_argument('--pretrained', action='store_true',
                        help='ImageNet initialisation from $CCST_PRETRAINED_DIR/<network>.pth (the reference downloads it, nets/resnet.py:364-369; '
                             'there is no network here, so it is opt-in and fails loudly when the file is missing)')
    parser.add_argument('--workers', type=int, default=0, help='DataLoader decode workers (the reference uses 0)')
    return parser.parse_args()


def restore(checkpoint, server_model, models, fedbn):
    """fed_run.py:626-640 (--resume) / :585-589 (--test): the server model from 'server_model'; every client from its own
    'model_{k}' under --mode fedbn (local BN weights and running statistics survive the restart), else from the server's.
    `models` maps client index -> model (one entry per rank under torchrun).  Returns the iteration to resume at."""
    server_model.load_state_dict(checkpoint['server_model'])
    for ci, m in models.items():
        m.load_state_dict(checkpoint['model_{}'.format(ci)] if fedbn else checkpoint['server_model'])
    return int(checkpoint['a_iter']) + 1


def main():
    args = parse()
    if args.tent_test or getattr(args, 'tent_test_on-the-fly', False) or getattr(args, 'tent_test_on_the_fly', False):
        raise NotImplementedError("--tent_test / --tent_test_on-the-fly (Tent test-time optimisation, fed_run.py:591-624) are outside the hot path")
    if args.mode.lower() not in ('fedavg', 'fedbn', 'deepall') or args.dg_method not in ('no_DG',):
        raise NotImplementedError("only --mode fedavg/fedbn/deepall with --dg_method no_DG is on the hot path")
    if not torch.cuda.is_available():
        raise SystemExit("ccst_amd: an MI355X (ROCm) device is required; there is no CPU path")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu)))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=device)
    seed = args.seed
    random.seed(a=seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    print('Device:', device)
    exp_folder = f'{args.dataset}/{args.mode}_{args.fusion_mode}_{args.dg_method}_{args.network}_locIter{args.wk_iters}/Target_{args.target}_seed_{seed}'
    args.save_path = os.path.join(args.save_path, exp_folder)
    SAVE_PATH = os.path.join(args.save_path, '{}'.format(args.mode))
    if rank == 0:
        os.makedirs(args.save_path, exist_ok=True)
    logfile = open(os.path.join(args.save_path, '{}.log'.format(args.mode)), 'a') if (args.log and rank == 0) else None

    print("Building server's model...")
    if not args.pretrained:
        print("NOTE: random (kaiming) initialisation; the reference starts from ImageNet weights -- pass --pretrained with "
              "CCST_PRETRAINED_DIR set to reproduce that")
    server_model = get_network(args.network)(args, pretrained=args.pretrained, classes=args.n_classes)
    loss_fun = fed.CrossEntropyLoss()
    print("Preparing data...")
    train_loaders, val_loaders, target_test_loader = data.get_fed_dataloaders(args, args.txt_root)
    datasets = args.source
    client_num = len(datasets) if args.mode != 'deepall' else 1
    client_weights = [float(1. / client_num) for _ in range(client_num)]
    if world > 1 and world != client_num:
        raise SystemExit("torchrun --nproc-per-node must equal the number of --source clients (%d)" % client_num)
    my_clients = [rank] if world > 1 else list(range(client_num))
    server_model.to(device)
    models = {ci: copy.deepcopy(server_model) for ci in my_clients}

    fedbn = args.mode.lower() == 'fedbn'
    if args.test:                                                    # fed_run.py:582-596
        print('Loading snapshots...')
        checkpoint = torch.load(SAVE_PATH, map_location='cpu')
        if fedbn and world > 1:
            raise SystemExit("--test of a fedbn checkpoint averages every client's BN entries: run it as one process")
        restore(checkpoint, server_model, models if fedbn else {}, fedbn)
        if fedbn:
            _, test_acc = fed.test_fedbn(server_model, [models[ci] for ci in my_clients], target_test_loader, loss_fun, device, args)
            print(' {:<11s}| Test  Acc: {:.4f}'.format(datasets[my_clients[-1]], test_acc))       # the reference prints the last client's name
        else:
            _, test_acc = fed.test(server_model, target_test_loader, loss_fun, device, args)
            print(' {:<11s}| Test  Acc: {:.4f}'.format(args.target, test_acc))
        return
    resume_iter = 0
    if args.resume:                                                  # fed_run.py:626-640
        checkpoint = torch.load(SAVE_PATH + '_latest', map_location='cpu')
        resume_iter = restore(checkpoint, server_model, models, fedbn)
        print('Resume training from epoch {}'.format(resume_iter))

    def log(msg):
        print(msg)
        if logfile:
            logfile.write(msg + "\n")
            logfile.flush()

    best_val_class_acc, best_test = 0., 0.
    for a_iter in range(resume_iter, args.iters):
        log("=============Global iter is {} ===============".format(a_iter))
        log("----------------Training----------------")
        optimizers = {ci: fed.SGD(models[ci], lr=args.lr) for ci in my_clients}      # fresh every round, :657
        for wi in range(args.wk_iters):
            iter_idx = wi + a_iter * args.wk_iters
            log("== Train epoch {} ===".format(iter_idx))
            for ci in my_clients:
                train_loss, train_acc = fed.train(models[ci], train_loaders[ci], optimizers[ci], loss_fun, client_num, device,
                                                  args, iter_idx, None)
                log(' {:<11s}| Train Loss: {:.4f}'.format(datasets[ci], train_loss))
                log(' {:<11s}| Train Class Acc: {:.4f}'.format(datasets[ci], train_acc))
        with torch.no_grad():
            if world > 1 and fedbn:
                fed.communication_distributed(args, models[rank], client_weights[rank], server_model=server_model)
                srv = server_model                    # the average lives in this rank's server replica
            elif world > 1:
                fed.communication_distributed(args, models[rank], client_weights[rank])
                srv = models[rank]                    # after the all-reduce every rank holds the server model
            else:
                server_model, ms = fed.communication(args, server_model, [models[ci] for ci in my_clients], client_weights)
                srv = server_model
            print("----------------Validate global model on source domains----------------")
            val_acc_sum = 0.0
            for ci in my_clients:
                # FedBN validates the local model on its source domain (fed_run.py:693-698)
                val_loss, val_acc = fed.test(models[ci] if fedbn else srv, val_loaders[ci], loss_fun, device, args)
                log(' {:<11s}| Global Val Loss: {:.4f}'.format(datasets[ci], val_loss))
                log(' {:<11s}| Global Val Class Acc: {:.4f}'.format(datasets[ci], val_acc))
                val_acc_sum += val_acc
            if world > 1:
                t = torch.tensor([val_acc_sum], device=device, dtype=torch.float64)
                dist.all_reduce(t)
                val_acc_sum = float(t)
            val_class_acc_average = val_acc_sum / client_num
            # does this round write a checkpoint?  val_class_acc_average is all-reduced, so every rank takes the same decision
            # without another collective (best_val_class_acc is tracked on every rank for that)
            will_save = (a_iter % args.save_freq == 0 and a_iter > 0) or val_class_acc_average > best_val_class_acc
            client_states = None
            if fedbn and world > 1 and will_save:
                # one client per rank: rank 0 collects every client's state for the checkpoint (:735-739) -- K x 94 MB over RCCL plus K
                # device-to-host copies, so only on the rounds that use it
                client_states = fed.gather_client_states(models[rank])
            if rank != 0 and val_class_acc_average > best_val_class_acc:
                best_val_class_acc = val_class_acc_average
            if rank == 0:
                print("-------------Test server model on target domain testset----------------")
                test_loss, test_acc = fed.test(srv, target_test_loader, loss_fun, device, args)
                log(' {:<11s}| Global Test Loss: {:.4f}'.format(args.target, test_loss))
                log(' {:<11s}| Global Test Class Acc: {:.4f}'.format(args.target, test_acc))
                ckpt = {'server_model': {k: v.detach().cpu() for k, v in srv.state_dict().items()}, 'a_iter': a_iter} if will_save else None
                if fedbn and will_save:       # :735-739
                    if client_states is not None:
                        for ci, sd in enumerate(client_states):
                            ckpt['model_{}'.format(ci)] = sd
                    else:
                        for ci in my_clients:
                            ckpt['model_{}'.format(ci)] = {k: v.detach().cpu() for k, v in models[ci].state_dict().items()}
                if a_iter % args.save_freq == 0 and a_iter > 0:
                    torch.save(ckpt, SAVE_PATH + '_latest')
                if val_class_acc_average > best_val_class_acc:
                    best_val_class_acc, best_test = val_class_acc_average, test_acc
                    log(' Saving current best checkpoints to {}...'.format(SAVE_PATH))
                    torch.save(ckpt, SAVE_PATH)
    if logfile:
        logfile.write(f'Test result using the global model with best val accuracy: {best_test} on {args.target}')
        logfile.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
