def rgb_to_grayscale(img, num_output_channels=1):
    r, g, b = img.unbind(dim=-3)
    l_img = (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)
    if num_output_channels == 3:
        return l_img.expand(img.shape)
    return l_img


def normalize(tensor, mean, std, inplace=False):
    import torch
    mean = torch.as_tensor(mean, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
    std = torch.as_tensor(std, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
    return (tensor - mean) / std
