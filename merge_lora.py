"""Merge a trained adapter into the base model and export it (reference merge_lora.py:13-55, same flags).

    python merge_lora.py --lora_model=output/.../checkpoint-final --model_path=<dir|synthetic:base> --eeg_ch=208

W <- W + scale * B A (LoRA) or W + scale/(r+1e-5) * B (A * E) (AdaLoRA) on every adapted Linear, the trained
front-end / conv2 copies (modules_to_save) replace the base ones, and the result is written to
<lora_model>/full_model as config.json + model.safetensors with HuggingFace names.  Unlike the stock loader, this
build's `from_pretrained` restores a saved MEG front-end (`model.encoder.conv1.0/2.*` or a stride-2
`model.encoder.conv1.*`), so `evaluation.py --model_path=<...>/full_model` needs no `--lora_model`.
"""
import argparse
import functools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from neuspeech1_amd.peft_compat import PeftModel  # noqa: E402
from utils.load_model import WhisperForConditionalGeneration  # noqa: E402
from utils.model_utils import projection_module  # noqa: E402
from utils.utils import add_arguments, print_arguments  # noqa: E402


# what WhisperFeatureExtractor / WhisperTokenizerFast / WhisperProcessor .save_pretrained write
PROCESSOR_FILES = ("preprocessor_config.json", "processor_config.json", "tokenizer.json", "tokenizer_config.json", "vocab.json",
                   "merges.txt", "normalizer.json", "added_tokens.json", "special_tokens_map.json", "generation_config.json")


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__)
    add_arg = functools.partial(add_arguments, argparser=parser)
    add_arg("lora_model", type=str, default="output/whisper-tiny/checkpoint-final/", help="adapter directory")
    add_arg("model_path", type=str, default="openai/whisper-base", help="base model")
    add_arg("eeg_ch", type=int, default=0, help="MEG channels of the front-end")
    add_arg("local_files_only", type=bool, default=True, help="never download")
    add_arg("config_name", type=str, default="base", help="front-end module ('base' | 'replace')")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    print_arguments(args)
    assert os.path.exists(args.lora_model), f"adapter {args.lora_model} not found"
    model = WhisperForConditionalGeneration.from_pretrained(args.model_path, device_map="auto",
                                                            local_files_only=args.local_files_only)
    conv1 = projection_module(config_name=args.config_name, meg_ch=args.eeg_ch,
                              d_model=model.model.encoder.conv2.in_channels).to(model.device)
    model.model.encoder.set_input_embeddings(conv1)
    model = PeftModel.from_pretrained(model, args.lora_model, local_files_only=args.local_files_only)
    model = model.merge_and_unload()
    model.train(False)
    save_directory = os.path.join(args.lora_model, "full_model")
    os.makedirs(save_directory, exist_ok=True)
    model.save_pretrained(save_directory)
    # the reference saves feature extractor, tokenizer and processor beside the merged weights (merge_lora.py:24-29,
    # 50-54) so that evaluation.py can load WhisperProcessor from full_model: their files travel unchanged
    base = args.model_path
    if isinstance(base, str) and os.path.isdir(base):
        import shutil
        for name in PROCESSOR_FILES:
            src = os.path.join(base, name)
            if os.path.isfile(src) and not os.path.exists(os.path.join(save_directory, name)):
                shutil.copy2(src, os.path.join(save_directory, name))
    with open(os.path.join(save_directory, "merge_info.json"), "w") as f:
        json.dump({"base_model": args.model_path, "adapter": os.path.abspath(args.lora_model), "eeg_ch": args.eeg_ch,
                   "config_name": args.config_name}, f, indent=1)
    print(f"merged model saved to {save_directory}")
    return save_directory


if __name__ == "__main__":
    main()
