"""Oracle (test infrastructure only): FedAvg aggregation and the train/test
loops of federated/fed_run.py on CPU."""
import torch


def communication_fedavg(server_model, models, client_weights):
    """federated/fed_run.py:400-414 (the branch every non-fedbn mode takes,
    SURVEY.md 3.5).  For each key: 'num_batches_tracked' -> server takes client
    0's value and the clients keep their own; otherwise server = sum_i w_i *
    client_i (accumulated in client order) and every client is overwritten."""
    with torch.no_grad():
        ssd = server_model.state_dict()
        for key in ssd.keys():
            if 'num_batches_tracked' in key:
                ssd[key].data.copy_(models[0].state_dict()[key])
            else:
                temp = torch.zeros_like(ssd[key])
                for ci in range(len(client_weights)):
                    temp += client_weights[ci] * models[ci].state_dict()[key]
                ssd[key].data.copy_(temp)
                for ci in range(len(client_weights)):
                    models[ci].state_dict()[key].data.copy_(ssd[key])
    return server_model, models


def communication_fedbn(server_model, models, client_weights):
    """federated/fed_run.py:388-399 (--mode fedbn): the server averages every key exactly as above,
    but clients are only overwritten for keys whose NAME does not contain 'bn' -- so bn1/bnN weights,
    biases and running stats stay local, while a downsample branch's BatchNorm ('downsample.1.*')
    is still shared."""
    with torch.no_grad():
        ssd = server_model.state_dict()
        for key in ssd.keys():
            if 'num_batches_tracked' in key:
                ssd[key].data.copy_(models[0].state_dict()[key])
            else:
                temp = torch.zeros_like(ssd[key])
                for ci in range(len(client_weights)):
                    temp += client_weights[ci] * models[ci].state_dict()[key]
                ssd[key].data.copy_(temp)
                if 'bn' not in key:
                    for ci in range(len(client_weights)):
                        models[ci].state_dict()[key].data.copy_(ssd[key])
    return server_model, models


def train_epoch(model, loader, lr, loss_fun, log=None):
    """federated/fed_run.py:31-88 without device moves: returns (train_loss, train_acc) =
    (sum loss / n_batches, correct / num_data).  `log`, when a list, receives what the reference hands its
    logger per iteration (:72-75): (loss, samples right, batch size).  Pinned bit-for-bit against the
    reference's own train() by tests/golden/fed_loop.npz (tools/make_golden.py)."""
    model.train()
    num_data, correct, loss_all, it = 0, 0, 0.0, -1
    for it, (img, lab) in enumerate(loader):
        for p in model.parameters():
            p.grad = None
        logit = model(img)
        loss = loss_fun(logit, lab)
        loss_all += loss.item()
        right = int((logit.max(dim=1)[1] == lab).sum())
        correct += right
        num_data += img.size(0)
        if log is not None:
            log.append((loss.item(), right, img.shape[0]))
        loss.backward()
        with torch.no_grad():
            for p in model.parameters():
                p.add_(p.grad, alpha=-lr)   # torch.optim.SGD._single_tensor_sgd form
    return loss_all / (it + 1), float(correct) / num_data


def test_epoch(model, loader, loss_fun):
    """federated/fed_run.py:214-259 (IN_test off): eval-mode forward.  Pinned by tests/golden/fed_loop.npz."""
    model.eval()
    num_data, correct, loss_all, it = 0, 0, 0.0, -1
    with torch.no_grad():
        for it, (img, lab) in enumerate(loader):
            logit = model(img)
            loss_all += loss_fun(logit, lab).item()
            correct += int((logit.max(dim=1)[1] == lab).sum())
            num_data += img.size(0)
    return loss_all / (it + 1), float(correct) / num_data


def test_fedbn_merge(server_model, models):
    """federated/fed_run.py:350-362, the state merge test_fedbn() does before its evaluation loop: the server takes client 0's
    num_batches_tracked and, for every OTHER key containing 'bn', the 1/K average of the clients' entries."""
    client_num = len(models)
    client_weights = [float(1. / client_num) for _ in range(client_num)]
    with torch.no_grad():
        for key in models[0].state_dict().keys():
            if 'num_batches_tracked' in key:
                server_model.state_dict()[key].data.copy_(models[0].state_dict()[key])
            if 'bn' in key and 'num_batches_tracked' not in key:
                temp = torch.zeros_like(server_model.state_dict()[key])
                for ci in range(client_num):
                    temp += client_weights[ci] * models[ci].state_dict()[key]
                server_model.state_dict()[key].data.copy_(temp)
    return server_model
