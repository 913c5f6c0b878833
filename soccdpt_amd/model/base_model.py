"""BaseModel: checkpoint loading / device lookup, same contract as
/root/reference/SOccDPT/model/base_model.py:5-46."""
import torch


class BaseModel(torch.nn.Module):
    def load_net(self, path):
        """Load a state dict from `path` (strict=False; mismatches are printed, not raised).
        `None` / `False` / "" mean "nothing to load"."""
        if path is None or not path:
            return
        parameters = torch.load(path, map_location=torch.device("cpu"))
        if "optimizer" in parameters:
            print("Loading optimizer state dict")
            parameters = parameters["model"]
        incompatible_keys = self.load_state_dict(parameters, strict=False)
        print("incompatible_keys", incompatible_keys)
        del parameters

    def get_device(self):
        try:
            return next(self.parameters()).device
        except Exception as ex:  # no parameters
            print("No device found, using CPU", ex)
            return torch.device("cpu")
