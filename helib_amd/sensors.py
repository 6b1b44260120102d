"""Engine clock and package power of one GPU while a loop runs (measurement hygiene, not part of the data path).

The fresh-multiply loop runs at the package power cap (1330-1400 W, 2.0-2.2 GHz: profiles/r04_power_and_clocks_under_load.txt)
and the pool's boxes differ by 15 % on one binary; a benchmark line that carries the clock and the power it was measured at
lets a reader tell a slow box from a regression.  `Sampler` reads the amdgpu hwmon files of the device (freq1_input = sclk in
Hz, power1_average / power1_input in microwatts) from a background thread -- a few file reads per sample, no process spawned
inside the timed region -- and falls back to `rocm-smi --json` when the files are not there."""
import glob
import json
import os
import subprocess
import threading
import time


def _hwmon_dirs():
    """[(pci address 'dddd:bb:dd.f', hwmon directory)] of every amdgpu device the kernel shows -- on a partitioned
    node that is all eight GPUs, whichever of them this process may use"""
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if "-" in os.path.basename(card):
            continue
        hw = sorted(glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")))
        if hw and os.path.exists(os.path.join(card, "device", "vendor")):
            try:
                if open(os.path.join(card, "device", "vendor")).read().strip() == "0x1002":
                    out.append((os.path.basename(os.path.realpath(os.path.join(card, "device"))).lower(), hw[0]))
            except OSError:
                pass
    return out


def pci_address(device=0):
    """PCI address of HIP device `device` as sysfs spells it, through torch when it is there (None otherwise)"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def _read_dir(d):
    hz = _read_int(os.path.join(d, "freq1_input"))
    uw = _read_int(os.path.join(d, "power1_average"))
    if uw is None:
        uw = _read_int(os.path.join(d, "power1_input"))
    if hz is None and uw is None:
        return None
    return {"sclk_mhz": None if hz is None else round(hz / 1e6), "power_w": None if uw is None else round(uw / 1e6, 1)}


def sample_sysfs(device=0, address=None):
    """the device's own sensors when its PCI address is known; otherwise the amdgpu device drawing the most power
    (on a node whose other GPUs idle that is the one under this process's load -- the source string says which rule)"""
    dirs = _hwmon_dirs()
    if not dirs:
        return None
    if address:
        for a, d in dirs:
            if a == address.lower():
                s = _read_dir(d)
                if s:
                    s["picked"] = "pci " + a
                return s
    best = None
    for a, d in dirs:
        s = _read_dir(d)
        if s and (best is None or (s["power_w"] or 0) > (best["power_w"] or 0)):
            best = dict(s, picked="highest power of %d amdgpu devices (%s)" % (len(dirs), a))
    return best


def sample_rocm_smi(device=0):
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--json"], capture_output=True,
                           text=True, timeout=10)
        d = json.loads(r.stdout)
        card = d.get(f"card{device}") or next(iter(d.values()))
        sclk = next((v for k, v in card.items() if k.lower().startswith("sclk")), None)
        power = next((v for k, v in card.items() if "power" in k.lower() and "w" in k.lower()), None)
        mhz = None
        if sclk:
            import re
            m = re.search(r"(\d+)\s*mhz", str(sclk), re.I)
            mhz = int(m.group(1)) if m else None
        return {"sclk_mhz": mhz, "power_w": None if power is None else float(power)}
    except Exception:
        return None


class Sampler:
    """with Sampler(device) as s: <timed loop>; s.summary() -> {"clock_mhz_under_load": [...], "power_w": [...], ...}"""

    def __init__(self, device=0, period_s=0.25, max_samples=64):
        self.device, self.period, self.max = device, period_s, max_samples
        self.samples, self.source = [], None
        self.address = pci_address(device)
        self._stop = threading.Event()
        self._thread = None

    def _one(self, allow_spawn=False):
        s = sample_sysfs(self.device, self.address)
        if s is not None:
            self.source = "amdgpu hwmon (freq1_input, power1_average), " + s.pop("picked", "")
            return s
        # no hwmon files on this box: rocm-smi is a process and a driver query -- never from inside the timed loop
        # (ADVICE r5); one sample as the window opens and one as it closes instead
        if not allow_spawn:
            return None
        s = sample_rocm_smi(self.device)
        if s is not None:
            self.source = "rocm-smi --showclocks --showpower --json, ONE sample before and one after the timed region (no hwmon files here)"
        return s

    def _run(self):
        while not self._stop.is_set() and len(self.samples) < self.max:
            s = self._one()
            if s is None:
                return
            s["t"] = time.perf_counter()
            self.samples.append(s)
            self._stop.wait(self.period)

    def __enter__(self):
        self._t0 = time.perf_counter()
        self._spawned = sample_sysfs(self.device, self.address) is None
        if self._spawned:                       # (before the timed loop starts)
            s = self._one(allow_spawn=True)
            if s is not None:
                s["t"] = time.perf_counter()
                self.samples.append(s)
            return self
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=15)
        self.window_s = time.perf_counter() - self._t0
        if self._spawned:                       # (after it has ended)
            s = self._one(allow_spawn=True)
            if s is not None:
                s["t"] = time.perf_counter()
                self.samples.append(s)
        return False

    def summary(self):
        # the first sample is taken as the loop starts (the queue is still filling): the figures are over the rest
        body = self.samples[1:] if len(self.samples) > 3 else self.samples
        clk = [s["sclk_mhz"] for s in body if s.get("sclk_mhz")]
        pw = [s["power_w"] for s in body if s.get("power_w")]
        if not clk and not pw:
            return {"clock_mhz_under_load": None, "power_w": None, "sensor_source": "unavailable on this box"}

        def three(v):   # first, middle, last of the window
            return [v[0], v[len(v) // 2], v[-1]] if v else None
        out = {"clock_mhz_under_load": three(clk), "power_w": three(pw),
               "clock_mhz_min_max": [min(clk), max(clk)] if clk else None,
               "power_w_min_max": [min(pw), max(pw)] if pw else None,
               "sensor_samples": len(body), "sensor_source": self.source,
               "sensor_mode": "two samples around the timed region" if getattr(self, "_spawned", False) else "sampled every %.2f s inside it" % self.period}
        # power1_average is the driver's own running average over about a second: in a window shorter than that it still
        # shows the load BEFORE the window (a 0.3 s leg reported 379 W three times) -- not a figure of this window
        if getattr(self, "window_s", 10.0) < 1.0 and not getattr(self, "_spawned", False):
            out["power_w"] = out["power_w_min_max"] = None
            out["power_note"] = "window of %.2f s: shorter than the sensor's averaging time, power not reported" % self.window_s
        return out
