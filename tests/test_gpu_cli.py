"""extract_feature.py CLI (-m gpu): flags and on-disk layout of the reference's output stage
(/root/reference/extract_feature.py:113-148, figures/output_format.jpg) on the synthetic front-end."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    (tmp_path / "imgs").mkdir()
    for n in ("a", "b", "c"):
        Image.fromarray((rs.rand(200, 240, 3) * 255).astype(np.uint8)).save(tmp_path / "imgs" / f"{n}.png")
    (tmp_path / "prompt.txt").write_text("a photo of a cat")
    layers = {"up-level1-repeat2-res-out": True, "up-level3-repeat0-vit-block0-self-k": True}
    (tmp_path / "layers.json").write_text(json.dumps(layers))
    return layers


def test_cli_layouts(tmp_path, monkeypatch):
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    sys.path.insert(0, ROOT)
    import extract_feature as cli
    layers = _setup(tmp_path)
    base = ["--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "256", "--t", "100", "-b", "2",
            "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt")]
    cli.main(base + ["--output_dir", str(tmp_path / "o1"), "--use_original_filename"])
    for k, (c, hw) in {"up-level1-repeat2-res-out": (1280, 8), "up-level3-repeat0-vit-block0-self-k": (320, 32)}.items():
        for n in ("a", "b", "c"):
            arr = np.load(tmp_path / "o1" / k / f"{n}.npy")
            assert arr.shape == (c, hw, hw) and arr.dtype == np.float16 and np.isfinite(arr.astype(np.float32)).all()
    cli.main(base + ["--output_dir", str(tmp_path / "o2"), "--sample_name_first", "--split", "val"])
    assert sorted(os.listdir(tmp_path / "o2")) == ["val0", "val1", "val2"]
    assert sorted(os.listdir(tmp_path / "o2" / "val1")) == sorted(k + ".npy" for k in layers)
    cli.main(base + ["--output_dir", str(tmp_path / "o3"), "--aggregate_output", "--use_original_filename"])
    agg = np.load(tmp_path / "o3" / "b.npy")
    assert agg.shape == (1280 + 320, 32, 32)
    # aggregated == nearest-resize + channel concat of the per-layer files of the same run configuration
    lo = np.load(tmp_path / "o1" / "up-level1-repeat2-res-out" / "b.npy")
    assert np.array_equal(agg[:1280, ::4, ::4], lo) and np.array_equal(agg[1280:], np.load(
        tmp_path / "o1" / "up-level3-repeat0-vit-block0-self-k" / "b.npy"))


def test_cli_precise_flag(tmp_path, monkeypatch):
    """--precise (native extension): the same files, written from split-operand plans — close to, not equal to, the default run's."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("GDF_PRECISE", "0")
    sys.path.insert(0, ROOT)
    import extract_feature as cli
    layers = _setup(tmp_path)
    base = ["--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "256", "--t", "100", "-b", "2",
            "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt"), "--use_original_filename"]
    cli.main(base + ["--output_dir", str(tmp_path / "d")])
    cli.main(base + ["--output_dir", str(tmp_path / "p"), "--precise"])
    assert os.environ.get("GDF_PRECISE") == "0"            # the flag travels as a constructor argument, not through the environment
    with pytest.raises(SystemExit):                        # the DiT versions have no split-operand plans: rejected, not silently ignored
        cli.main(["--layer", str(tmp_path / "layers.json"), "--version", "flux", "--input_dir", "x", "--prompt_file", str(tmp_path / "prompt.txt"),
                  "--output_dir", str(tmp_path / "f"), "--precise"])
    for k in layers:
        a = np.load(tmp_path / "d" / k / "a.npy").astype(np.float32); b = np.load(tmp_path / "p" / k / "a.npy").astype(np.float32)
        e = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert a.shape == b.shape and 0.0 < e < 2.5e-3, (k, e)


def test_cli_early_exit_flag_same_files(tmp_path, monkeypatch):
    """--early_exit (native extension): the forward stops after the last requested layer; the files are bit-identical."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    sys.path.insert(0, ROOT)
    import extract_feature as cli
    layers = _setup(tmp_path)
    base = ["--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "256", "--t", "100", "-b", "2",
            "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt"), "--use_original_filename"]
    cli.main(base + ["--output_dir", str(tmp_path / "full")])
    cli.main(base + ["--output_dir", str(tmp_path / "ee"), "--early_exit"])
    for k in layers:
        for n in "abc":
            assert np.array_equal(np.load(tmp_path / "full" / k / f"{n}.npy").view(np.uint16), np.load(tmp_path / "ee" / k / f"{n}.npy").view(np.uint16)), (k, n)


def test_output_stage_matches_reference_files_on_gpu(tmp_path):
    """(f)2 on device tensors: nearest resize + concat on the GPU, pinned async D2H, byte-identical to the reference's files."""
    from test_host_cpu import _run_output_stage, check_output_stage
    got, want = _run_output_stage(tmp_path, "cuda")
    check_output_stage(got, want)


def test_cli_loader_threads_equal_the_serial_input_loop(tmp_path, monkeypatch):
    """Round 5: the CLI decodes / resizes / normalises the next batches on loader threads (extract_feature.BatchLoader) while the GPU works; the
    threads run FeatureExtractor.preprocess_image — the serial path's own function — so `--loader_threads 0` (the reference's serial loop) and the
    default give byte-identical files.  Five images, batch 2: a ragged last batch and more batches than the prefetch depth."""
    monkeypatch.setenv("GDF_SYNTHETIC_WEIGHTS", "1")
    sys.path.insert(0, ROOT)
    import extract_feature as cli
    from PIL import Image
    layers = _setup(tmp_path)
    rs = np.random.RandomState(3)
    for n in ("d", "e"):
        Image.fromarray((rs.rand(150, 130, 3) * 255).astype(np.uint8)).save(tmp_path / "imgs" / f"{n}.png")
    base = ["--layer", str(tmp_path / "layers.json"), "--version", "1-5", "--img_size", "256", "--t", "100", "-b", "2",
            "--input_dir", str(tmp_path / "imgs" / "*.png"), "--prompt_file", str(tmp_path / "prompt.txt")]
    cli.main(base + ["--output_dir", str(tmp_path / "serial"), "--loader_threads", "0"])
    cli.main(base + ["--output_dir", str(tmp_path / "threads")])
    cli.main(base + ["--output_dir", str(tmp_path / "one"), "--loader_threads", "1"])
    for k in layers:
        for i in range(5):
            a = np.load(tmp_path / "serial" / k / f"train{i}.npy")
            for other in ("threads", "one"):
                b = np.load(tmp_path / other / k / f"train{i}.npy")
                assert a.dtype == b.dtype and np.array_equal(a.view(np.uint16), b.view(np.uint16)), (k, i, other)
    # a file that cannot be decoded surfaces as an error of the batch it belongs to, not as a hang
    (tmp_path / "imgs" / "zz_broken.png").write_bytes(b"not an image")
    with pytest.raises(Exception):
        cli.main(base + ["--output_dir", str(tmp_path / "bad")])
