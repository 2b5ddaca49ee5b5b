"""Shader clock and package power under load (rocm-smi, read-only)."""


class ClockSampler:
    """Reads the GPU's shader clock and package power (rocm-smi, read-only; a child process every ~0.2 s from a thread of rank 0) while the
    timed regions run: the roofline figures assume 2.4 GHz, the chip decides what it sustains under this instruction mix."""

    def __init__(self, device):
        import threading
        self.device, self.samples, self._stop = device, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _once(self):
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.device), "--showclocks", "--showpower", "--csv"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5).stdout
            rows = [r for r in out.splitlines() if r.strip()]
            d = dict(zip(rows[0].split(","), rows[1].split(",")))
            sclk = int(re.sub(r"[^0-9]", "", d.get("sclk clock speed:", "")) or 0)
            pw = float(d.get("Current Socket Graphics Package Power (W)", "0") or 0)
            if sclk > 0:
                self.samples.append((sclk, pw))
        except Exception:
            pass

    def _run(self):
        while not self._stop.is_set():
            self._once()
            self._stop.wait(0.2)

    def start(self):
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        self._th.join(timeout=10)
        if not self.samples:
            return None
        sc = [x[0] for x in self.samples]; pw = [x[1] for x in self.samples]
        return {"sclk_mhz_mean": sum(sc) / len(sc), "sclk_mhz_min": min(sc), "sclk_mhz_max": max(sc), "package_power_w_mean": sum(pw) / len(pw), "samples": len(sc),
                "how": "rocm-smi --showclocks --showpower every ~0.2 s over the HBM-resident timed region (rank 0's GPU; the host is idle there)",
                "note": "every roofline / issue-floor figure of this line assumes 2.4 GHz; the chip sustains what its power management allows for the instruction mix"}
