"""Device-pointer entry points of the C ABI (pvw_encrypt_device back to back,
pvw_encrypt_multi_device, pvw_decrypt_noisy_device, pvw_decode_device) against the host-buffer paths."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_device_pointer_entry_points():
    out = subprocess.run([sys.executable, os.path.join(HERE, "_device_api_worker.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DEVICE_API_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
