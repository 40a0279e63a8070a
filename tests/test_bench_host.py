"""Host-side helpers of bench.py that need no GPU: the board-power sensor lookup (amdgpu hwmon files:
power1_average or power1_input in microwatts, power1_cap) on a made-up sysfs tree."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _card(root, name, pci, files):
    dev = os.path.join(root, "devices", pci)
    hw = os.path.join(dev, "hwmon", "hwmon7")
    os.makedirs(hw)
    for f, v in files.items():
        with open(os.path.join(hw, f), "w") as fh:
            fh.write(v)
    os.makedirs(os.path.join(root, "drm", name))
    os.symlink(dev, os.path.join(root, "drm", name, "device"))
    return hw


def _torch(bus):
    props = types.SimpleNamespace(pci_bus_id=bus)
    return types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda i: props))


def test_no_cards_means_no_power_record(tmp_path):
    os.makedirs(tmp_path / "drm")
    assert bench.board_power_sensor(_torch(0), 0, root=str(tmp_path / "drm")) is None


def test_single_card_is_taken(tmp_path):
    hw = _card(str(tmp_path), "card0", "0000:05:00.0", {"power1_input": "1400000000\n", "power1_cap": "1400000000\n"})
    power, cap = bench.board_power_sensor(_torch(99), 0, root=str(tmp_path / "drm"))
    assert power == os.path.join(str(tmp_path), "drm", "card0", "device", "hwmon", "hwmon7", "power1_input")
    assert bench.read_number(power) == 1.4e9 and bench.read_number(cap) == 1.4e9
    assert bench.read_number(os.path.join(hw, "missing")) is None


def test_card_is_matched_by_pci_bus(tmp_path):
    _card(str(tmp_path), "card0", "0000:05:00.0", {"power1_average": "1\n"})
    _card(str(tmp_path), "card1", "0000:c5:00.0", {"power1_average": "2\n"})
    power, _ = bench.board_power_sensor(_torch(0xC5), 1, root=str(tmp_path / "drm"))
    assert bench.read_number(power) == 2.0
    assert bench.board_power_sensor(_torch(0x11), 0, root=str(tmp_path / "drm")) is None
