"""The loader side of the training driver: ``get_data_loader`` with the reference's settings (reference
data/__init__.py:32-52: shuffle, drop_last, pinned) and a synthetic image dataset (no dataset files on the MI355X box;
BASELINE configs use synthetic Gaussian images).  Decoding/augmentation of real datasets is outside the accelerated
path; uint8 batches go through ``functional.images_from_uint8`` on the device."""
import torch
from torch.utils.data import DataLoader, Dataset


def get_data_loader(dataset, batch_size, num_workers):
    return DataLoader(dataset, batch_size=batch_size, shuffle=True, num_workers=num_workers, drop_last=True, pin_memory=True)


class SyntheticImages(Dataset):
    """``num_images`` fixed N(0,1) images [channels, resolution, resolution] fp32, generated from ``seed`` on demand."""

    def __init__(self, num_images, resolution, channels=3, seed=0):
        self.n, self.res, self.ch, self.seed = int(num_images), int(resolution), int(channels), int(seed)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator()
        g.manual_seed(self.seed * 1000003 + int(i))
        return torch.randn(self.ch, self.res, self.res, generator=g)
