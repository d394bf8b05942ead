"""Reference import path ``utils.load_dataset`` -> lstc_vad_amd.load_dataset (same class and function names)."""
from lstc_vad_amd.load_dataset import (  # noqa: F401
    SH_Train_Origin_Dataset, SH_Train_Origin_Dataset_MutualTraining, SH_Train_Origin_Dataset_tenCrop,
    UBnormal_Train_Origin_Dataset, UBnormal_Train_Origin_Dataset_tenCrop, UCF_Train_Origin_Dataset,
    UBnormal_test, UBnormal_test_tenCrop, UCF_test, UCF_test_tenCrop, UCF_train, shanghaitech_test,
    shanghaitech_test_tenCrop, ResidentPairs,
)
