"""Schedule / target-copy helpers with the reference's names (torchrl/algo/utils.py:23-32)."""


def copy_model_params_from_to(source, target):
    for dst, src in zip(target.parameters(), source.parameters()):
        dst.data.copy_(src.data)


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """lr = lr0 * (1 - epoch / total), evaluated in the reference's operation order."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for group in optimizer.param_groups:
        group["lr"] = lr
